"""Register-level model of the pipelined cores' clocking (test infrastructure).

Tracks, clock by clock, WHAT sits in each of the L = NSTAGES+2 registers of
rtl/cordic.v / rtl/topolar.v -- either the input sample of some earlier clock
or a register cleared by i_reset -- and evaluates the output register with the
oracle.  Deliberately written as a shift register of tokens (not as the
prefix-scan formulation the GPU uses) so that the two can disagree."""
import numpy as np

import oracle_lib as O


class PipeModel:
    def __init__(self, ocfg, rot):
        self.ocfg, self.rot = ocfg, rot
        self.ns = ocfg.nstages
        self.L = self.ns + 2
        live = [ocfg.angle[i] if (ocfg.angle[i] != 0 and i < ocfg.ww) else 0
                for i in range(self.ns)]
        self.live = live
        self.samples = []                 # (x, y, phase, aux) ever accepted
        self.pipe = [("z", j) for j in range(self.L)]   # power-up == reset

    def run(self, x, y, ph, ce=None, rs=None, aux=None):
        n = len(x)
        o0 = np.zeros(n, dtype=np.int64)
        o1 = np.zeros(n, dtype=np.int64)
        oa = np.zeros(n, dtype=np.uint8)
        want = []                         # (clock, sample index) to evaluate
        pmask = (1 << self.ocfg.pw) - 1
        for t in range(n):
            if rs is not None and rs[t]:
                self.pipe = [("z", j) for j in range(self.L)]
            elif ce is None or ce[t]:
                self.samples.append((int(x[t]), int(y[t]),
                                     int(ph[t]) if self.rot else 0,
                                     int(aux[t]) if aux is not None else 0))
                self.pipe = [("s", len(self.samples) - 1)] + self.pipe[:-1]
            kind, v = self.pipe[-1]
            if kind == "s":
                want.append((t, v))
                oa[t] = 1 if self.samples[v][3] else 0
            elif not self.rot and v <= self.ns:
                # a cleared phase register that has passed stages v .. NSTAGES-1
                o1[t] = sum(self.live[v:self.ns]) & pmask
        if want:
            idx = [v for _, v in want]
            sx = np.array([self.samples[i][0] for i in idx], dtype=np.int32)
            sy = np.array([self.samples[i][1] for i in idx], dtype=np.int32)
            if self.rot:
                sp = np.array([self.samples[i][2] for i in idx], dtype=np.uint32)
                a, b = O.rotate(self.ocfg, sx, sy, sp)
            else:
                a, b = O.topolar(self.ocfg, sx, sy)
            ts = [t for t, _ in want]
            o0[ts] = a
            o1[ts] = b.astype(np.int64) if not self.rot else b
        return o0, o1, oa
