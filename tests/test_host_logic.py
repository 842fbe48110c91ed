"""Host-side arithmetic of the training path that needs no GPU: the split chooser of the grouped weight-gradient launch, the workspace size the C-ABI
reports for it, and the workgroup count of the LayerNorm backward (the sizes the Python layer and a C caller must agree on)."""
import ctypes as C

import pytest


def test_wgrad_group_split_chooser_fills_whole_rounds():
    from multimodal_amd import ops

    # a ViT-B/16 layer at B = 256: 27 + 9 + 36 + 36 tiles over 788 K-tiles -> 7 splits = 756 workgroups = 2.95 rounds of 256
    assert ops.wgrad_group_splits(108, 50432 // 64) == 7
    # the text layer (12 + 4 + 16 + 16 tiles, 308 K-tiles): 5 splits = 240 workgroups, one round
    assert ops.wgrad_group_splits(48, 19712 // 64) == 5
    # ViT-L/14 (192 tiles): 4 splits = exactly three rounds
    assert ops.wgrad_group_splits(192, 32896 // 64) == 4
    # never fewer than 32 K-tiles per workgroup, never less than one split
    assert ops.wgrad_group_splits(4, 64) <= 2 and ops.wgrad_group_splits(1000, 2) == 1
    for tiles in (1, 7, 100, 255, 256, 257, 1000):
        for kt in (2, 40, 400, 4000):
            s = ops.wgrad_group_splits(tiles, kt)
            assert 1 <= s <= max(1, kt // 32)


def test_grouped_wgrad_workspace_size_and_argument_checks():
    from multimodal_amd import _lib, ops

    L = _lib.lib()
    jobs = (ops._WgradJob * 3)()
    for j, (M, N, K, db) in zip(jobs, ((768, 3072, 50432, 1), (768, 768, 50432, 0), (512, 2048, 256, 1))):
        j.M, j.N, j.K, j.db = M, N, K, db  # (db: any non-NULL value means "wanted" for the size query)
    # K = 50432 at 7 splits: 788 K-tiles -> chunks of 114 (even) -> 7 parts; K = 256 at 7 splits: 4 K-tiles -> chunks of 2 -> 2 parts
    want = 4 + 7 * 768 * 3072 + 7 * 768 + 7 * 768 * 768 + 2 * 512 * 2048 + 2 * 512
    assert L.mmamd_gemm_bf16_tn_splitk_group_ws(C.cast(jobs, C.c_void_p), 3, 7) == want
    # one split: every problem writes its own result, no partials
    assert L.mmamd_gemm_bf16_tn_splitk_group_ws(C.cast(jobs, C.c_void_p), 3, 1) == 4
    jobs[2].K = 200  # not a multiple of 128
    assert L.mmamd_gemm_bf16_tn_splitk_group_ws(C.cast(jobs, C.c_void_p), 3, 7) == -1
    assert L.mmamd_gemm_bf16_tn_splitk_group_ws(None, 0, 7) == -1
    # argument errors of the launch itself are reported before anything touches a device
    assert L.mmamd_gemm_bf16_tn_splitk_group(None, 0, 1, None, None) != 0
    assert L.mmamd_gemm_bf16_tn_splitk_group(C.cast(jobs, C.c_void_p), 9, 1, None, None) != 0


@pytest.mark.parametrize("rows,d,want", [(50432, 768, 1024), (50432, 1024, 768), (19712, 512, 768), (1000, 768, 250), (3, 2048, 1), (4096, 640, 1024), (5000, 512, 768)])
def test_layernorm_backward_workgroup_count(rows, d, want):
    from multimodal_amd import _lib

    assert _lib.lib().mmamd_layernorm_bwd_groups(rows, d) == want


def test_step_timeline_tool_on_a_synthetic_trace(tmp_path, capsys):
    """tools/step_timeline.py (busy / idle / co-running time per queue from a rocprofv3 kernel trace) on a hand-made trace: two queues, one 10 us gap."""
    import sys

    from tools import step_timeline

    rows = ['"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp"']
    ev = [(1, "gemm", 0, 100_000), (2, "ln", 50_000, 80_000), (1, "attn", 110_000, 200_000), (2, "ln", 120_000, 150_000)]  # ns
    for i, (q, name, s, e) in enumerate(ev):
        rows.append(f'"KERNEL_DISPATCH","Agent 2",{q},0,1,{i},1,"{name}",{i},{1_000_000 + s},{1_000_000 + e}')
    f = tmp_path / "t_kernel_trace.csv"
    f.write_text("\n".join(rows) + "\n")
    argv = sys.argv
    try:
        sys.argv = ["step_timeline.py", str(f), "--last-ms", "1"]
        step_timeline.main()
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    assert "4 kernels" in out and "q1: 0.2 ms" in out and "q2: 0.1 ms" in out
    # 200 us window: idle 10 us (5 %), one kernel 130 us, two kernels 60 us
    assert "0.0 ms (5.0 %) / 0.1 ms (65.0 %) / 0.1 ms (30.0 %)" in out
    assert "1 idle gaps" in out and "after gemm" in out and "before attn" in out


def test_persistent_gemm_tile_order_visits_every_tile_once():
    """The tile order of the persistent GEMM kernels (column chunk outermost, then groups of gm row tiles, then the chunk's column tiles: csrc/gemm.hip
    tile_order_map, the SAME function the kernels call) enumerated on the host through mmamd_debug_tile_order: a bijection onto the tile grid for the
    launchers' own (gm, cn) choices and for forced ones -- including ragged last chunks, last row groups, single-row / single-column grids and the 193-column
    vocabulary GEMM of CoCa.  No GPU."""
    import ctypes as C

    import numpy as np

    from multimodal_amd import _lib

    L = _lib.lib()
    cases = [(197, 12, 768, 0, 0), (197, 9, 768, 0, 0), (197, 3, 3072, 0, 0), (77, 8, 512, 0, 0), (257, 16, 1024, 0, 0), (38, 193, 768, 0, 0),
             (5, 7, 4096, 0, 0), (1, 1, 64, 0, 0), (1, 13, 2048, 0, 0), (13, 1, 2048, 0, 0)]
    cases += [(tm, tn, 768, gm, cn) for tm in (1, 2, 7, 31) for tn in (1, 2, 5, 12) for gm in (1, 2, 3, 8) for cn in (1, 2, 5, 6, 12, 20)]
    for tiles_m, tiles_n, K, gm, cn in cases:
        out = np.full((tiles_m * tiles_n, 2), -1, dtype=np.int32)
        used = L.mmamd_debug_tile_order(tiles_m, tiles_n, K, gm, cn, out.ctypes.data_as(C.c_void_p))
        assert 1 <= used <= tiles_n, (tiles_m, tiles_n, K, gm, cn, used)
        assert out[:, 0].min() >= 0 and out[:, 0].max() == tiles_m - 1 and out[:, 1].min() >= 0 and out[:, 1].max() == tiles_n - 1
        flat = out[:, 0].astype(np.int64) * tiles_n + out[:, 1]
        assert len(np.unique(flat)) == tiles_m * tiles_n, (tiles_m, tiles_n, K, gm, cn)
        # chunks are contiguous in the order: the column-chunk index never decreases along the list
        ck = out[:, 1] // used
        assert (np.diff(ck) >= 0).all(), (tiles_m, tiles_n, K, gm, cn)
    # the launchers' choices: wide GEMMs with a W above 3 MB are chunked (MLP-up 12 -> 6 + 6, qkv 9 -> 5 + 4), few-column / small-W ones are not
    pick = lambda tn, K: L.mmamd_debug_tile_order(4, tn, K, 0, 0, np.zeros((4 * tn, 2), dtype=np.int32).ctypes.data_as(C.c_void_p))  # noqa: E731
    assert pick(12, 768) == 6 and pick(9, 768) == 5 and pick(3, 3072) == 3 and pick(8, 512) == 8 and pick(3, 768) == 3


def test_whole_row_gemm_layernorm_shape_gate_and_argument_checks():
    """mmamd_gemm_bf16_residual_ln_supported / _grouped: the shape gate the schedules would consult, and the argument errors, without a GPU."""
    from multimodal_amd import _lib

    L = _lib.lib()
    ok = L.mmamd_gemm_bf16_residual_ln_supported
    assert ok(50432, 768, 768) == 1 and ok(19712, 512, 512) == 1 and ok(1, 768, 64 * 3) == 1
    assert ok(128, 1024, 1024) == 0      # N: 512 or 768 only (a stage of W must fit a 48 KiB ring slot)
    assert ok(128, 768, 48) == 0         # K: a multiple of 32, at least three stages
    assert ok(0, 768, 768) == 0
    assert ok(800000, 768, 768) == 0     # M N 4 bytes must stay below 2^31 (buffer-descriptor range)
    assert L.mmamd_gemm_bf16_residual_ln_grouped(None, 1, None) == _lib.E_BADARG if hasattr(_lib, "E_BADARG") else L.mmamd_gemm_bf16_residual_ln_grouped(None, 1, None) < 0
    assert L.mmamd_pack_w_ksteps(None, 768, 768, None, None) < 0
    assert b"pack_w_ksteps" in L.mmamd_last_error()
