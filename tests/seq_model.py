"""Clock-level model of the sequential cores' handshake (test infrastructure).

rtl/seqcordic.v:226-327 / rtl/seqpolar.v:211-307 reduced to what a bench can
observe when it keeps to the protocol: with C = CLOCKS_PER_OUTPUT,
  * a sample is accepted on a clock with i_stb while the core is idle;
  * o_busy reads 1 after that clock and the next C-2;
  * the clock C-1 after the accept registers the result: o_done reads 1 for
    that one clock, o_xval/o_yval (o_mag/o_phase) and o_aux change;
  * i_stb while busy is ignored -- EXCEPT on the completing clock itself,
    where the RTL keeps `idle` low and runs the datapath again over its own
    result: this closed-form model only counts such a clock (`violations`);
    the register-level model that reproduces it is oracle_lib.seq_trace
    (orc_seq_trace), which is what the GPU is checked against off protocol;
  * i_reset (wins over i_stb) drops a sample in flight, clears o_done and the
    aux register, and leaves the output registers alone -- which still load
    on a reset that hits the completing clock.
"""
import numpy as np

import oracle_lib as O


class SeqModel:
    def __init__(self, ocfg, rot):
        self.ocfg, self.rot = ocfg, rot
        self.C = ocfg.clocks_per_output
        self.c = 0                       # clocks left until the result loads
        self.pending = None              # (x, y, phase, aux) in flight
        self.out = (0, 0, 0)             # output registers
        self.violations = 0

    def _eval(self, s):
        x = np.array([s[0]], dtype=np.int32)
        y = np.array([s[1]], dtype=np.int32)
        if self.rot:
            a, b = O.rotate(self.ocfg, x, y, np.array([s[2]], dtype=np.uint32))
        else:
            a, b = O.topolar(self.ocfg, x, y)
        return int(a[0]), int(b[0])

    def run(self, stb, x, y, ph=None, rs=None, aux=None):
        n = len(stb)
        o0 = np.zeros(n, dtype=np.int64)
        o1 = np.zeros(n, dtype=np.int64)
        oa = np.zeros(n, dtype=np.uint8)
        busy = np.zeros(n, dtype=np.uint8)
        done = np.zeros(n, dtype=np.uint8)
        for t in range(n):
            rst = rs is not None and rs[t]
            completing = self.c == 1
            if completing:               # output registers load, reset or not
                a, b = self._eval(self.pending)
                self.out = (a, b, self.pending[3])
            if rst:
                self.c = 0
            elif self.c == 0:
                if stb[t]:
                    self.pending = (int(x[t]), int(y[t]),
                                    int(ph[t]) if self.rot else 0,
                                    int(aux[t]) if aux is not None else 0)
                    self.c = self.C - 1
            else:
                if completing:
                    done[t] = 1
                    if stb[t]:
                        self.violations += 1
                self.c -= 1
            o0[t], o1[t], oa[t] = self.out
            busy[t] = 1 if self.c > 0 else 0
        return o0, o1, oa, busy, done
