"""The N>1 path on CPU: two gloo ranks shard a batch by global sample index,
each computes its block (the oracle stands in for the device kernel -- this
test is about the sharding / digest / gather logic, cordic_amd/shard.py),
and the reduced digest and the gathered outputs must equal the unsharded run."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from cordic_amd.shard import chunk_ranges, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1000, (1 << 33) + 5):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert spans[-1][0] + spans[-1][1] == n
            sizes = [c for _, c in spans]
            assert max(sizes) - min(sizes) <= 1


def test_c_abi_shard_range_matches():
    """cordic_shard_range (C++, what cordic_group uses) == shard.py."""
    import cordic_amd as ca
    for n in (0, 1, 7, 8, 1000, (1 << 33) + 5):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert ca.shard_range(n, r, world) == shard_range(n, r, world)
    import pytest
    with pytest.raises(ca.CordicError):
        ca.shard_range(10, 2, 2)


def test_chunk_ranges_cover_without_overlap():
    for n in (0, 1, 5, 8, 4099, 1 << 30):
        for k in (1, 3, 8, 16):
            pieces = chunk_ranges(n, k)
            assert [a for a, _ in pieces[1:]] == [b for _, b in pieces[:-1]]
            assert (pieces[0][0], pieces[-1][1]) == (0, n) if n else not pieces
            assert all(b > a for a, b in pieces)


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_lib as O
    from gpu_util import cpu_digest
    from cordic_amd.shard import (shard_range, reduce_digest, gather_to_root,
                                  pipelined_gather)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N = 100003
    cfg = O.config_cli(O.P2R, 32, 32, 2, 32, 24)     # BASELINE config 4 core
    start, cnt = shard_range(N, rank, world)
    import cordic_amd as ca
    assert ca.shard_range(N, rank, world) == (start, cnt)   # the C ABI's split
    g = np.arange(cnt, dtype=np.uint64) + np.uint64(start)
    ph = (g & np.uint64(0xffffffff)).astype(np.uint32)   # phase[n] = (uint32)n
    ox, oy = O.rotate(cfg, 2**31 - 1, 0, ph)
    local = (cpu_digest(ox, start) + cpu_digest(oy, start + (1 << 40))) %% 2**64
    total = reduce_digest(local)
    gx = gather_to_root(torch.from_numpy(ox), N)
    gy = gather_to_root(torch.from_numpy(oy), N)
    # equal shards computed and collected chunk by chunk (the bench's --gather)
    M = 4099
    gph = ((np.arange(M, dtype=np.uint64) + np.uint64(rank * M))
           & np.uint64(0xffffffff)).astype(np.uint32)
    px, py = torch.zeros(M, dtype=torch.int32), torch.zeros(M, dtype=torch.int32)
    def compute(a, b):
        cx, cy = O.rotate(cfg, 2**31 - 1, 0, gph[a:b])
        px[a:b] = torch.from_numpy(cx); py[a:b] = torch.from_numpy(cy)
    got = pipelined_gather(compute, [px, py], chunks=7)
    if rank == 0:
        allph = (np.arange(world * M, dtype=np.uint64) & np.uint64(0xffffffff)).astype(np.uint32)
        wx, wy = O.rotate(cfg, 2**31 - 1, 0, allph)
        assert np.array_equal(torch.cat(got[0]).numpy(), wx)
        assert np.array_equal(torch.cat(got[1]).numpy(), wy)
    else:
        assert got is None
    if rank == 0:
        ph_all = (np.arange(N, dtype=np.uint64) & np.uint64(0xffffffff)).astype(np.uint32)
        rx, ry = O.rotate(cfg, 2**31 - 1, 0, ph_all)
        want = (cpu_digest(rx, 0) + cpu_digest(ry, 1 << 40)) %% 2**64
        assert total == want, (hex(total), hex(want))
        assert np.array_equal(gx.numpy(), rx) and np.array_equal(gy.numpy(), ry)
        print("SHARD-OK")
    else:
        assert gx is None and gy is None
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARD-OK" in outs[0]
