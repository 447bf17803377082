"""CPU, 2 and 8 processes, gloo: the camera-shard / all-gather / un-pad logic of far3d_amd.dist (no compute kernels run)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from far3d_amd import dist as fdist


def test_camera_shards_layout():
    assert fdist.camera_shards(7, 1) == (7, [[0, 1, 2, 3, 4, 5, 6]])
    assert fdist.camera_shards(7, 2) == (4, [[0, 1, 2, 3], [4, 5, 6, -1]])
    assert fdist.camera_shards(7, 4) == (2, [[0, 1], [2, 3], [4, 5], [6, -1]])
    per, sh = fdist.camera_shards(7, 8)
    assert per == 1 and sh[6] == [6] and sh[7] == [-1]
    for w in (1, 2, 3, 4, 8):
        per, sh = fdist.camera_shards(7, w)
        flat = [c for s in sh for c in s]
        assert [c for c in flat if c >= 0] == list(range(7)) and len(flat) == per * w and all(len(s) == per for s in sh)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    num_cams, S, C, K = 7, 5, 4, 3
    per, shards = fdist.camera_shards(num_cams, world)
    full_tok = torch.arange(num_cams * S * C, dtype=torch.float32).view(num_cams, S, C)
    full_rec = torch.arange(num_cams * K * 6, dtype=torch.float32).view(num_cams, K, 6) * 0.5
    mine = [c for c in shards[rank] if c >= 0]
    tok = torch.cat([full_tok[mine], torch.full((per - len(mine), S, C), -7.0)])      # garbage in the padding slot
    rec = torch.cat([full_rec[mine], torch.full((per - len(mine), K, 6), -7.0)])
    work, tok_fn = fdist.gather_camera_major(tok, num_cams, async_op=True)
    _, rec_fn = fdist.gather_camera_major(rec, num_cams)
    work.wait()
    ok = torch.equal(tok_fn(), full_tok) and torch.equal(rec_fn(), full_rec)
    # persistent destination (what ShardedFrame hands the replicated head): gathered in place, twice
    dst = torch.full((world * per, S, C), -1.0)
    for _ in range(2):
        _, fn = fdist.gather_camera_major(tok, num_cams, out=dst)
        ok = ok and fn().data_ptr() == dst.data_ptr() and torch.equal(dst[:num_cams], full_tok)
    # the frame's ONE exchange: value maps, records and per-rank counts in one call (coalesced on RCCL, sequential on gloo);
    # fixed-capacity blocks: rank r contributes a block of 4 rows with r + 2 valid ones -> (world, 4, 6) + counts
    blk = torch.full((4, 6), float(rank + 1))
    cnt = torch.tensor([[rank + 2, 0]], dtype=torch.int32)
    d_tok, d_blk, d_cnt = torch.empty((world * per, S, C)), torch.empty((world * 4, 6)), torch.empty((world, 2), dtype=torch.int32)
    for w in fdist.gather_many([(tok, d_tok), (blk, d_blk), (cnt, d_cnt)]):
        w.wait()
    ok = ok and torch.equal(d_tok[:num_cams], full_tok) and d_cnt[:, 0].tolist() == [r + 2 for r in range(world)]
    ok = ok and all(bool((d_blk[4 * r:4 * r + 4] == r + 1).all()) for r in range(world))
    # what far3d_compact_rows (HIP, ops.compact_rows) does with the gathered blocks in ShardedFrame._head: the first cnt[r] rows of every
    # rank's block, in rank order -- rank order IS camera order (contiguous camera blocks per rank), so the packed rows are the
    # reference's camera-major proposal order; an idle rank (no camera: 8 ranks, 7 cameras) contributes a zero count and no row.
    # Here each rank's block carries its cameras' proposals tagged (camera, index); counts differ per camera.
    ncam_props = [c % 3 + 1 for c in range(num_cams)]                            # proposals per camera
    rows = [(c, j) for c in mine for j in range(ncam_props[c])]
    cap = per * 3
    blk2 = torch.full((cap, 2), -9.0)
    if rows:
        blk2[:len(rows)] = torch.tensor(rows, dtype=torch.float32)
    cnt2 = torch.tensor([[len(rows), 0]], dtype=torch.int32)
    g_blk, g_cnt = torch.empty((world * cap, 2)), torch.empty((world, 2), dtype=torch.int32)
    for w in fdist.gather_many([(blk2, g_blk), (cnt2, g_cnt)]):
        w.wait()
    packed = torch.cat([g_blk.view(world, cap, 2)[r, :int(g_cnt[r, 0])] for r in range(world)])
    want = torch.tensor([(c, j) for c in range(num_cams) for j in range(ncam_props[c])], dtype=torch.float32)
    ok = ok and torch.equal(packed, want) and int(g_cnt[:, 0].sum()) == sum(ncam_props)
    if world == 8:
        ok = ok and (len(mine) == 0) == (rank == 7) and int(g_cnt[7, 0]) == 0
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]


def test_gather_camera_major_two_ranks_gloo():
    _run_world(2)


def test_gather_eight_ranks_seven_cameras_one_idle_rank_gloo():
    """BASELINE configs[2]'s layout: 8 ranks, 7 cameras -- rank 7 owns only a padding slot, contributes padding to the value maps, a
    zero count to the fixed-capacity records, and the packed proposal rows still come out in the reference's camera-major order."""
    _run_world(8)
