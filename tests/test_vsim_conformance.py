"""Conformance of tests/vsim.py -- the Verilog-subset interpreter that stands
behind the per-sample vectors -- with IEEE 1364-2005.  Each case is a few lines
of Verilog whose result is worked out BY HAND from the LRM rule it names
(5.4.1 expression bit lengths, 5.5.1 expression types, 5.1.12 shift operators,
5.1.14 concatenations, 9.2.2 non-blocking assignments); none of the expected
values comes out of a simulator or out of vsim itself.

vsim.py is this project's own reading of Verilog, not Verilator: these tests
narrow what that reading can get wrong, they do not make it a reference build.
"""
import pytest

import vsim

HEAD = """
module t(input wire i_clk, output reg [15:0] o);
"""


def run(decls, body, ticks=1, comb=""):
    m = vsim.Module(HEAD + decls + comb +
                    "\n    always @(posedge i_clk) begin\n" + body +
                    "\n    end\nendmodule\n")
    for _ in range(ticks):
        m.tick()
    return m


def u(m, name):
    return m.get(name)


# ---- 5.5.1: an expression is signed only if ALL its operands are

def test_mixed_signedness_is_unsigned():
    m = run("""
    reg signed [3:0] a; reg [3:0] b; reg signed [3:0] c;
    reg [7:0] r_mixed; reg [7:0] r_signed; reg signed [7:0] r_sdst;
    initial a = -1; initial b = 1; initial c = 1;
    """, """
        r_mixed  <= a + b;      // b unsigned: a is ZERO-extended -> 15 + 1
        r_signed <= a + c;      // both signed: -1 + 1 in 8 bits
        r_sdst   <= a + b;      // a signed target changes nothing
    """)
    assert u(m, "r_mixed") == 16
    assert u(m, "r_signed") == 0
    assert u(m, "r_sdst") == 16


def test_signed_operands_are_sign_extended_to_the_context():
    m = run("""
    reg signed [3:0] a; reg signed [7:0] r8; reg [7:0] ru;
    initial a = -3;
    """, """
        r8 <= a;                // 4'sb1101 -> 8'b1111_1101
        ru <= a;                // assignment to an unsigned reg still extends
                                // the signed RHS by its sign (5.5.4)
    """)
    assert u(m, "r8") == 0xFD and u(m, "ru") == 0xFD


def test_literal_signedness():
    m = run("""
    reg signed [7:0] r1, r2, r3; reg signed [3:0] a; initial a = -2;
    """, """
        r1 <= a + 4'sd1;        // signed literal: -2 + 1 = -1
        r2 <= a + 4'd1;         // unsigned literal: 14 + 1 = 15
        r3 <= a + 1;            // unsized decimal is signed
    """)
    assert (u(m, "r1"), u(m, "r2"), u(m, "r3")) == (0xFF, 15, 0xFF)


# ---- 5.4.1: the context is as wide as the wider side of the assignment

def test_carry_is_kept_only_if_the_context_has_room():
    m = run("""
    reg [3:0] a, b; reg [3:0] r4; reg [4:0] r5; reg [3:0] rs;
    initial a = 15; initial b = 1;
    """, """
        r5 <= a + b;            // 5-bit context: 16
        r4 <= a + b;            // 4-bit context: carry lost
        rs <= (a + b) >> 1;     // still a 4-bit context: (0) >> 1, not 8
    """)
    assert (u(m, "r5"), u(m, "r4"), u(m, "rs")) == (16, 0, 0)


def test_wider_target_widens_the_operands_before_the_operation():
    m = run("""
    reg [3:0] a; reg [7:0] rn, rm; initial a = 4'b0101;
    """, """
        rn <= ~a;               // a extended to 8 bits FIRST: ~0000_0101
        rm <= -a;               // 8-bit two's complement of 5
    """)
    assert (u(m, "rn"), u(m, "rm")) == (0xFA, 0xFB)


def test_wire_sum_then_part_select():
    # what the generator does for rounding: a wider wire, then a part select
    m = run("""
    reg [3:0] a, b; reg [3:0] r;
    wire [4:0] w;
    assign w = a + b;
    initial a = 15; initial b = 3;
    """, """
        r <= w[4:1];            // (18) >> 1 through the 5-bit wire
    """)
    assert u(m, "w") == 18 and u(m, "r") == 9


# ---- 5.1.12: >>> is arithmetic only if the result type is signed

def test_arithmetic_shift_needs_a_signed_context():
    m = run("""
    reg signed [7:0] s; reg [7:0] v; reg signed [7:0] r1; reg [7:0] r2, r3, r4;
    initial s = 8'hF0; initial v = 8'hF0;
    """, """
        r1 <= s >>> 2;          // signed: -16 >>> 2 = -4
        r2 <= v >>> 2;          // unsigned operand: logical
        r3 <= (s >>> 2) + v;    // v makes the WHOLE expression unsigned:
                                // 8'hF0 >>> 2 is logical (3C) + F0 = 12C -> 2C
        r4 <= s >> 2;           // >> is always logical
    """)
    assert u(m, "r1") == 0xFC and u(m, "r2") == 0x3C
    assert u(m, "r3") == 0x2C and u(m, "r4") == 0x3C


def test_shift_amount_is_unsigned_and_self_determined():
    m = run("""
    reg signed [7:0] s; reg [2:0] k; reg signed [7:0] r; initial s = -128;
    initial k = 3'd7;
    """, """
        r <= s >>> k;           // -128 >>> 7 = -1
    """)
    assert u(m, "r") == 0xFF


# ---- 5.1.14: concatenation / replication are unsigned, widths add up

def test_concatenation_replication_and_signed_cast():
    m = run("""
    reg [3:0] a; reg signed [7:0] r1, r2; reg [7:0] r3; reg [5:0] r4;
    initial a = 4'b1010;
    """, """
        r1 <= {a[3], a[2:0]};           // a concatenation is UNSIGNED: 0000_1010
        r2 <= $signed({a[3], a[2:0]});  // 4-bit signed -6 -> 1111_1010
        r3 <= {2{a}};                   // 1010_1010
        r4 <= {{2{a[3]}}, a};           // sign extension idiom: 11_1010
    """)
    assert (u(m, "r1"), u(m, "r2"), u(m, "r3"), u(m, "r4")) == (
        0x0A, 0xFA, 0xAA, 0x3A)


def test_concatenation_operands_are_self_determined():
    m = run("""
    reg [3:0] a, b; reg [7:0] r; initial a = 15; initial b = 1;
    """, """
        r <= {a + b, 4'h0};     // inside {}: a 4-bit sum, the carry is lost
    """)
    assert u(m, "r") == 0x00


# ---- relational operators: signed only if both sides are

def test_comparisons():
    m = run("""
    reg signed [3:0] a; reg [3:0] b; reg signed [3:0] c;
    reg r1, r2, r3; initial a = -1; initial b = 1; initial c = 1;
    """, """
        r1 <= (a < c);          // signed: -1 < 1
        r2 <= (a < b);          // unsigned: 15 < 1
        r3 <= (a == 4'hF);      // bit pattern equality
    """)
    assert (u(m, "r1"), u(m, "r2"), u(m, "r3")) == (1, 0, 1)


def test_comparison_extends_the_narrower_operand():
    m = run("""
    reg signed [3:0] a; reg signed [7:0] w; reg r; initial a = -1;
    initial w = -1;
    """, """
        r <= (a == w);          // both signed: a sign-extended to 8 bits
    """)
    assert u(m, "r") == 1


# ---- 9.2.2: non-blocking assignments sample before any of them updates

def test_non_blocking_ordering():
    m = run("""
    reg [7:0] a, b, c; initial a = 1; initial b = 2; initial c = 0;
    """, """
        a <= b;                 // a swap
        b <= a;
        c <= c + 1;             // two assignments to one reg: the last wins,
        c <= c + 2;             // both read the OLD c
    """, ticks=1)
    assert (u(m, "a"), u(m, "b"), u(m, "c")) == (2, 1, 2)
    m.tick()
    assert (u(m, "a"), u(m, "b"), u(m, "c")) == (1, 2, 4)


def test_pipeline_registers_shift_by_one_per_clock():
    m = run("""
    reg [7:0] s0, s1, s2; initial s0 = 7; initial s1 = 0; initial s2 = 0;
    """, """
        s1 <= s0;
        s2 <= s1;
        s0 <= s0 + 1;
    """, ticks=2)
    assert (u(m, "s0"), u(m, "s1"), u(m, "s2")) == (9, 8, 7)


# ---- the multiplication of the quadratic-interpolation core: signed x signed

def test_signed_multiply_width():
    m = run("""
    reg signed [7:0] a; reg signed [7:0] b; reg signed [15:0] p; reg [15:0] q;
    reg [7:0] ub;
    initial a = -3; initial b = 5; initial ub = 5;
    """, """
        p <= a * b;             // 16-bit context, signed: -15
        q <= a * ub;            // unsigned: 253 * 5 (8-bit operands zero-
                                // extended to 16 bits) = 1265
    """)
    assert u(m, "p") == 0xFFF1 and u(m, "q") == 1265


def test_generate_for_and_bit_select():
    m = vsim.Module("""
module t(input wire i_clk, output reg [7:0] o);
    reg [7:0] v [0:3];
    genvar i;
    generate for(i=0; i<3; i=i+1) begin : G
        always @(posedge i_clk)
            v[i+1] <= v[i] + 1;
    end endgenerate
    initial v[0] = 10;
    always @(posedge i_clk)
        o <= { 7'h0, v[3][1] };
endmodule
""")
    for _ in range(4):
        m.tick()
    assert m.state["v"][1:] == [11, 12, 13]
    assert m.get("o") == ((13 >> 1) & 1)
