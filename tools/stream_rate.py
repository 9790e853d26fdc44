"""Rate of the clocked view (cordic_stream_ticks) on one GPU: clocks per second
for a block of 2^26 clocks of the cfg2 core, with and without per-clock
i_ce / i_reset arrays.  Prints one JSON line per case."""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cordic_amd as ca  # noqa: E402


def main():
    dev = "cuda:0"
    n = 1 << 26
    for name, cli in (("cfg2 p2r", (ca.P2R, 32, 32, 2, 32, 16)),
                      ("cfg3 r2p", (ca.R2P, 24, 24, 2, -1, 20))):
        cfg = ca.Config.from_cli(*cli)
        rot = cli[0] == ca.P2R
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.empty(n, dtype=torch.int32, device=dev)
        y = torch.empty(n, dtype=torch.int32, device=dev)
        ph = torch.empty(n, dtype=torch.int32, device=dev)
        lim = 1 << (cfg.iw - 1)
        x.random_(-lim, lim - 1, generator=g)
        y.random_(-lim, lim - 1, generator=g)
        ph.random_(-2**31, 2**31 - 1, generator=g)
        ce = (torch.rand(n, device=dev, generator=g) < 0.75).to(torch.uint8)
        rs = (torch.rand(n, device=dev, generator=g) < 1e-5).to(torch.uint8)
        aux = (torch.rand(n, device=dev, generator=g) < 0.5).to(torch.uint8)
        o0 = torch.empty(n, dtype=torch.int32, device=dev)
        o1 = torch.empty(n, dtype=torch.int32, device=dev)
        oa = torch.empty(n, dtype=torch.uint8, device=dev)
        s = ca.Stream(cfg)
        s.reserve(n)
        for label, kw in (("every clock enabled", {}),
                          ("i_ce 75 %, i_reset 1e-5, i_aux", dict(ce=ce, reset=rs, aux=aux))):
            for _ in range(2):
                s.ticks(x, y, ph if rot else None, o0, o1, oa, **kw)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            k = 10
            for _ in range(k):
                s.ticks(x, y, ph if rot else None, o0, o1, oa, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / k
            print(json.dumps({"core": name, "activity": label, "clocks": n,
                              "ms_per_block": ms,
                              "Mclocks_per_s": n / ms / 1e3}))


def seq_rate():
    dev = "cuda:0"
    n = 1 << 26
    for name, cli in (("cfg5 sp2r", (ca.SP2R, 32, 32, 2, 32, 16)),
                      ("seq cfg3 sr2p", (ca.SR2P, 24, 24, 2, -1, 20))):
        cfg = ca.Config.from_cli(*cli)
        rot = cli[0] == ca.SP2R
        g = torch.Generator(device=dev).manual_seed(2)
        x = torch.empty(n, dtype=torch.int32, device=dev)
        y = torch.empty(n, dtype=torch.int32, device=dev)
        ph = torch.empty(n, dtype=torch.int32, device=dev)
        lim = 1 << (cfg.iw - 1)
        x.random_(-lim, lim - 1, generator=g)
        y.random_(-lim, lim - 1, generator=g)
        ph.random_(-2**31, 2**31 - 1, generator=g)
        o0 = torch.empty(n, dtype=torch.int32, device=dev)
        o1 = torch.empty(n, dtype=torch.int32, device=dev)
        ob, od, oa = (torch.empty(n, dtype=torch.uint8, device=dev)
                      for _ in range(3))
        for label, p in (("i_stb every clock (1 result per %d clocks)"
                          % cfg.c.clocks_per_output, 1.0),
                         ("i_stb on 2 % of the clocks", 0.02)):
            stb = (torch.rand(n, device=dev, generator=g) < p).to(torch.uint8)
            s = ca.Seq(cfg)
            for _ in range(2):
                s.ticks(stb, x, y, ph if rot else None, o0, o1, ob, od, oa)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            k = 10
            for _ in range(k):
                s.ticks(stb, x, y, ph if rot else None, o0, o1, ob, od, oa)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / k
            print(json.dumps({"core": name, "activity": label, "clocks": n,
                              "ms_per_block": ms,
                              "Mclocks_per_s": n / ms / 1e3,
                              "results_per_block": int(od.sum().item())}))


if __name__ == "__main__":
    seq_rate()
    main()
