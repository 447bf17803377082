"""CPU: the measured tile table (far3d_amd/data/tuning_mi355x.json) must only name tiles the dispatcher of
far3d_conv2d_nhwc implements for that kind of layer -- a stale id would only surface as a runtime error on the GPU box."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dispatch_ids():
    src = open(os.path.join(ROOT, "far3d_amd", "csrc", "igemm.hip")).read()
    ids = {"igemm": set(), "dma": set(), "patch": set(), "pipe3": set(), "gemm": set()}
    for m in re.finditer(r"case (\d+): (?:rc = |return )?(launch_[a-z0-9_]+)<", src):
        tile, fn = int(m.group(1)), m.group(2)
        key = {"launch_igemm": "igemm", "launch_igemm_dma": "dma", "launch_conv3x3_patch": "patch",
               "launch_conv3x3_pipe": "pipe3", "launch_gemm1x1_pipe": "gemm", "launch_gemm1x1_wide": "gemm", "launch_gemm1x1_split": "gemm"}[fn]
        ids[key].add(tile)
    return ids


import pytest


@pytest.mark.parametrize("name", ["tuning_mi355x.json", "tuning_mi355x_tput.json"])      # tuned alone / under the frame pipeline's 3-stream concurrency
def test_tuning_table_names_only_implemented_tiles(name):
    ids = _dispatch_ids()
    assert ids["pipe3"] and ids["gemm"] and ids["dma"], ids
    table = json.load(open(os.path.join(ROOT, "far3d_amd", "data", name)))
    assert table, "empty tuning table"
    assert set(table) == set(json.load(open(os.path.join(ROOT, "far3d_amd", "data", "tuning_mi355x.json")))), "the two tables must cover the same layers"
    for key, tile in table.items():
        cout, cin, k, stride, npix = (int(v) for v in key.split(","))
        assert cout > 0 and cin > 0 and npix > 0 and k in (1, 3) and stride in (1, 2), key
        generic = ids["igemm"] | ids["dma"]                      # any kernel size / stride (bf16, Cin % 32 == 0 for the DMA ring)
        if k == 3 and stride == 1:
            allowed = generic | ids["patch"] | {t for t in ids["pipe3"] if not 30 <= t <= 39}
        elif k == 1 and stride == 1:
            allowed = generic | ids["gemm"]
        elif k == 3 and stride == 2:
            allowed = generic | {t for t in ids["pipe3"] if 30 <= t <= 39}
        else:
            allowed = generic
        assert tile in allowed, "tile %d is not dispatchable for layer %s" % (tile, key)
        if tile in ids["dma"] | ids["patch"] | ids["pipe3"] | ids["gemm"]:
            assert cin % 32 == 0, "LDS-DMA tile %d needs Cin %% 32 == 0 (%s)" % (tile, key)


def test_tuned_tile_lookup_borrows_nearest_pixel_count():
    """The table holds the same layer at several pixel counts (7 cameras on one GPU; 4 / 2 / 1 cameras per rank of a camera-sharded
    run): an exact pixel count gets its own entry, anything else the entry with the nearest pixel count."""
    from far3d_amd import ops
    table = json.load(open(os.path.join(ROOT, "far3d_amd", "data", "tuning_mi355x.json")))
    entries = sorted((int(k.split(",")[-1]), v) for k, v in table.items() if k.startswith("192,192,3,1,"))
    assert len(entries) >= 2, "the stage-4 3x3 layer should be tuned for the sharded pixel counts too"
    for npix, tile in entries:
        assert ops._tuned_tile(192, 192, 3, 1, npix) == tile
    lo, hi = entries[0], entries[-1]
    assert ops._tuned_tile(192, 192, 3, 1, lo[0] // 3) == lo[1] and ops._tuned_tile(192, 192, 3, 1, hi[0] * 3) == hi[1]
    assert ops._tuned_tile(191, 192, 3, 1, hi[0]) == 0                        # unknown layer -> kernel heuristic


@pytest.mark.parametrize("name", ["tuning_mi355x_pair.json", "tuning_mi355x_pair_tput.json"])
def test_pair_tuning_table_names_only_implemented_tiles(name):
    """Same for the pair-storage (bf16x3) table: ids 150+ of igemm_pair.hip, 1..5 for the register-staged kernel."""
    src = open(os.path.join(ROOT, "far3d_amd", "csrc", "igemm_pair.hip")).read()
    ids = {"igemm": set(), "pipe3": set(), "gemm": set()}
    for m in re.finditer(r"case (\d+): (?:rc = |return )?(launch_[a-z0-9_]+)<", src):
        ids[{"launch_igemm": "igemm", "launch_conv3x3_pipe": "pipe3", "launch_gemm1x1_pipe": "gemm"}[m.group(2)]].add(int(m.group(1)))
    ws_src = open(os.path.join(ROOT, "far3d_amd", "csrc", "conv_ws.hip")).read()
    # launch_conv3x3_ws<WGM, WGN, WM, WN, NP, PAIR, ...>: the tiles whose sixth template argument says pair storage
    ws_pair = {int(m.group(1)) for m in re.finditer(r"case (\d+): return launch_conv3x3_ws<([^>]*)>", ws_src) if m.group(2).split(",")[5].strip() == "true"}
    ws_gemm = {int(m.group(1)) for m in re.finditer(r"case (\d+): return launch_gemm1x1_ws<", ws_src)}      # persistent 1x1 GEMM (pair storage only)
    table = json.load(open(os.path.join(ROOT, "far3d_amd", "data", name)))
    assert table and ids["pipe3"] and ids["gemm"] and ids["igemm"] == {1, 2, 3, 4, 5} and ws_pair and all(400 <= t < 460 for t in ws_pair)
    assert ws_gemm and all(460 <= t < 478 for t in ws_gemm)
    assert set(table) == set(json.load(open(os.path.join(ROOT, "far3d_amd", "data", "tuning_mi355x_pair.json"))))
    from far3d_amd import ops
    for key, tile in table.items():
        cout, cin, k, stride, npix = (int(v) for v in key.split(","))
        assert cin % 32 == 0, key
        allowed = ids["igemm"] | ({t for t in ids["pipe3"] if t < 300} if (k == 3 and stride == 1) else ids["gemm"] if (k == 1 and stride == 1) else
                                  {t for t in ids["pipe3"] if t >= 330} if (k == 3 and stride == 2) else set())
        if isinstance(tile, list):
            # [wave-specialised tile (csrc/conv_ws.hip: 3x3 layers without residual / sums, 1x1 layers on pair maps), the general tile every other call of the layer shape takes]
            ws, tile = tile
            assert stride == 1 and cout % 32 == 0 and ws in ops.WS_TILES and ((k == 3 and ws in ws_pair) or (k == 1 and ws in ws_gemm)), (key, ws)
            assert ops._tuned_tile(cout, cin, k, stride, npix, name, ws_ok=True) == ws and ops._tuned_tile(cout, cin, k, stride, npix, name) == tile
        assert tile in allowed and not 200 <= tile < 300, "tile %d is not a split-product tile for layer %s" % (tile, key)


def test_persistent_tiles_are_not_borrowed_across_pixel_counts():
    """A [ws tile, general tile] entry hands out its persistent tile only near the pixel count it was measured at: a camera-sharded rank
    (1 of 7 cameras) borrows the entry of the 7-camera layer and must get the general tile."""
    from far3d_amd import ops
    name = "tuning_mi355x_pair.json"
    ws, gen = json.load(open(os.path.join(ROOT, "far3d_amd", "data", name)))["256,768,1,1,268800"]
    assert ws in ops.WS_TILES and gen not in ops.WS_TILES
    assert ops._tuned_tile(256, 768, 1, 1, 268800, name, ws_ok=True) == ws and ops._tuned_tile(256, 768, 1, 1, 268800, name) == gen
    assert ops._tuned_tile(256, 768, 1, 1, 230400, name, ws_ok=True) == ws          # 6 of 7 cameras: within a quarter
    for npix in (38400, 76800, 153600):                                             # 1, 2, 4 cameras per rank
        assert ops._tuned_tile(256, 768, 1, 1, npix, name, ws_ok=True) in (gen, 185), npix      # (185: the layer's own 16 800-pixel entry)


def test_tile_table_selection_is_scoped_and_thread_local():
    """ops.use_tile_tables: an engine's tables hold for the block only (nested selections restore each other), another thread keeps the
    process defaults meanwhile, and a custom bf16 table never becomes the pair table (the tables were re-assigned module
    globals, and the pair name was derived with str.replace)."""
    import threading
    from far3d_amd import ops
    base = ops.tile_tables()
    assert base == ("tuning_mi355x.json", "tuning_mi355x_pair.json")
    seen = {}

    def other():
        seen["other"] = ops.tile_tables()
    with ops.use_tile_tables("tuning_mi355x_tput.json"):
        assert ops.tile_tables() == ("tuning_mi355x_tput.json", "tuning_mi355x_pair_tput.json")
        t = threading.Thread(target=other)
        t.start()
        t.join()
        with ops.use_tile_tables("my_custom_table.json"):
            assert ops.tile_tables() == ("my_custom_table.json", "tuning_mi355x_pair.json")
        assert ops.tile_tables()[0] == "tuning_mi355x_tput.json"
        # the lookup follows the selection: stage-2 concat layer, latency table vs throughput table
        assert ops._tuned_tile(256, 768, 1, 1, 268800) == ops._tuned_tile(256, 768, 1, 1, 268800, "tuning_mi355x_tput.json")
    assert seen["other"] == base and ops.tile_tables() == base
    assert ops._tuned_tile(256, 768, 1, 1, 268800) == ops._tuned_tile(256, 768, 1, 1, 268800, "tuning_mi355x.json")


def test_engine_entry_points_run_under_the_engines_tables():
    import inspect
    from far3d_amd import engine
    for name in ("backbone", "fpn", "roi_head", "camera_stage", "head_stage", "forward_frame"):
        fn = getattr(engine.Far3DEngine, name)
        assert hasattr(fn, "__wrapped__") and "use_tile_tables" in inspect.getsource(engine._with_tile_tables), name


def test_serial_load_scan_counts_drained_loads():
    """tools/scan_serial_loads.py: a load followed by `s_waitcnt vmcnt(0)` before the next load counts, loads issued back to back before one
    wait do not, LDS-DMA loads are not looked at."""
    import sys
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scan_serial_loads as ssl
    serial = "\n".join("\tglobal_load_dword v%d, v[2:3], off\n\ts_waitcnt vmcnt(0)\n\tv_add_u32 v9, v9, v%d" % (i, i) for i in range(6))
    batched = "\n".join("\tglobal_load_dword v%d, v[2:3], off" % i for i in range(6)) + "\n\ts_waitcnt vmcnt(0)"
    dma = "\n".join("\tbuffer_load_dword v1, s[4:7], 0 offen lds\n\ts_waitcnt vmcnt(0)" for _ in range(6))
    txt = ""
    for name, body in (("_Z6serialv", serial), ("_Z7batchedv", batched), ("_Z3dmav", dma)):
        txt += "%s:                                ; @%s\n%s\n\ts_endpgm\n" % (name, name, body)
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(txt)
    rows = ssl.scan(f.name)
    os.unlink(f.name)
    assert [(r[1], r[2], r[3]) for r in rows] == [("_Z6serialv", 6, 6)]
