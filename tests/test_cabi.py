"""The C-ABI shared library loads and exports exactly what include/difformer_hip.h declares.
No compute call is made (there is no GPU here): only loading, host-side size arithmetic and the
argument checks that run before any HIP call."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "difformer_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dif_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from difformer_amd import _lib
    return _lib.load()


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    assert "dif_simple_reduce_f32" in names and "dif_gcn_spmm_f32" in names and "dif_gcn_spmm_tail_f32" in names and "dif_project_reduce_f32" in names and "dif_linear_f32" in names and "dif_gcn_spmm_tail_bf16" in names and "dif_rowgemm_f32" in names and "dif_subgraph" in names and "dif_batched_simple_attn_f32" in names and "dif_row_order" in names and "dif_gcn_spmm_part_f32" in names and "dif_sliced_spmm_f32" in names and "dif_simple_layer_f32" in names and "dif_subgraph_batches_group" in names and "dif_graph_prepare" in names and "dif_gcn_edge_weight_grad_f32" in names and "dif_batched_sigmoid_attn_bwd_f32" in names and "dif_tiny_forward_f32" in names and "dif_tiny_backward_f32" in names and "dif_tiny_graph_build" in names and "dif_set_exact_fp32" in names and len(names) == 104


def test_library_exports_every_declared_symbol(lib):
    raw = ctypes.CDLL(lib._name)
    for name in declared_functions():
        assert hasattr(raw, name), f"{name} declared in difformer_hip.h but not exported"


def test_python_binding_covers_the_header_exactly():
    from difformer_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_exact_fp32_switch_is_a_runtime_setting_of_the_library(lib):
    """dif_set_exact_fp32 returns what was set before: the launchers read the switch per call (host state, no device work)."""
    was = lib.dif_set_exact_fp32(1)
    try:
        assert lib.dif_set_exact_fp32(0) == 1 and lib.dif_set_exact_fp32(0) == 0
    finally:
        lib.dif_set_exact_fp32(was)


def test_version_and_size_helpers(lib):
    assert lib.dif_version() == 2
    assert lib.dif_simple_reduced_len(1, 64, 64) == 64 * 64 + 64 + 64 + 2          # 4,226 floats
    assert lib.dif_simple_reduced_len(2, 16, 16) == 2 * (256 + 32) + 2
    assert lib.dif_simple_workspace_bytes(132534, 1, 64, 64) >= 4226 * 4
    assert lib.dif_simple_workspace_bytes(0, 1, 64, 64) == 0
    small, big = lib.dif_csr_workspace_bytes(1000, 100, 1), lib.dif_csr_workspace_bytes(79255038, 132534, 13)
    assert 0 < small < big and big > 4 * 4 * 79255038
    assert lib.dif_sigmoid_workspace_bytes(16384, 16384, 1, 32, 32) == 0         # 512 query groups = one full round: no key split
    assert lib.dif_sigmoid_workspace_bytes(16384, 16384, 1, 64, 64) > 0          # 33 .. 64 columns from 2^25 pairs: the packed planes of csrc/sigmoid_wide.hip
    assert lib.dif_sigmoid_workspace_bytes(5000, 5000, 1, 64, 64) < lib.dif_sigmoid_workspace_bytes(6000, 6000, 1, 64, 64)
    assert lib.dif_sigmoid_workspace_bytes(20000, 20000, 1, 64, 64) > 0          # 625 groups: split to trim the last round
    assert lib.dif_sigmoid_workspace_bytes(2708, 2708, 1, 64, 64) > 0            # Cora: keys split over workgroups


def test_sliced_plan_is_host_side_arithmetic(lib):
    """dif_sliced_plan: geometry of the feature-sliced product (no device call)."""
    plan = (ctypes.c_int32 * 8)()
    assert lib.dif_sliced_plan(132534, 132534, 64, plan) == 0
    slices, panels, G, PW, W, R, T, NT = list(plan)
    assert (slices, panels) == (16, 16) and G == -(-132534 // 64) and PW == panels * W and (R - 1) * PW < G <= R * PW and W <= 16
    assert T % 16 == 0 and T <= 10208 and T * NT >= 132534 and NT == 13
    assert lib.dif_sliced_plan(2000000, 2000000, 64, plan) == 0 and plan[1] * plan[0] > 256     # more panels than CUs
    assert plan[5] <= 10
    assert lib.dif_sliced_plan(1000, 1000, 30, plan) == -2 and b"F % 4" in lib.dif_last_error()
    assert lib.dif_sliced_plan(5000, 5000, 64, plan) == 0 and plan[7] == 1                     # one tile: plain CSR
    rc = lib.dif_sliced_spmm_f32(None, None, plan, None, None, None, None, None, 5000, 5000, 0, 5000, 64, None, 0, 1.0, 1.0, None, 64, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.dif_last_error()
    rc = lib.dif_sliced_spmm_f32(None, None, plan, None, None, None, None, None, 5000, 6000, 0, 5000, 64, None, 0, 1.0, 1.0, None, 64, None, 0, None)
    assert rc == -1 and b"plan does not match" in lib.dif_last_error()
    # a row shard (1/8 of the rows over all the sources): 2 full panels, 8 source splits of 2 tiles each, and a workspace
    assert lib.dif_sliced_plan(132534, 16568, 64, plan) == 0
    slices, panels, G, PW, W, R, T, NT = list(plan)
    assert (slices, panels, R, NT) == (16, 2, 9, 16) and T * NT >= 132534 and R * PW >= G
    assert lib.dif_sliced_spmm_workspace_bytes(132534, 16568, 64) == 8 * 16 * G * 64 * 16
    assert lib.dif_sliced_spmm_workspace_bytes(132534, 132534, 64) == 0
    rc = lib.dif_sliced_spmm_f32(None, None, plan, None, None, None, None, None, 16568, 132534, 0, 16568, 64, None, 0, 1.0, 1.0, None, 64, None, 0, None)
    assert rc == -1 and b"workspace" in lib.dif_last_error()
    assert lib.dif_sliced_plan(132534, 66272, 64, plan) == 0 and (plan[1], plan[7]) == (8, 14)   # world 2: 2 splits
    # row positions: without `parts` there are exactly n_rows of them; with `parts` the order is required
    rc = lib.dif_sliced_spmm_f32(None, None, plan, None, None, None, None, None, 5064, 5000, 0, 5000, 64, None, 0, 1.0, 1.0, None, 64, None, 0, None)
    assert rc == -1 and b"n_pos" in lib.dif_last_error()


def test_sliced_plan_invariants_over_random_shapes(lib):
    """Every plan dif_sliced_plan hands out is self-consistent: the (panel, wave, round) slots cover the 64-row slots, the
    tiles cover the source rows and fit the LDS, a row shard's tiles divide evenly among its source splits, and the
    workspace is what the splits write."""
    import random
    rnd = random.Random(5)
    plan = (ctypes.c_int32 * 8)()
    for _ in range(3000):
        n_src = rnd.choice([rnd.randint(1, 5000), rnd.randint(5000, 300000), rnd.randint(300000, 4000000)])
        n_rows = n_src if rnd.random() < 0.4 else rnd.randint(1, n_src)
        F = 4 * rnd.choice([1, 2, 4, 8, 16, 16, 16, 32, 64, 75, 100])
        rc = lib.dif_sliced_plan(n_src, n_rows, F, plan)
        slices, panels, G, PW, W, R, T, NT = list(plan)
        if rc != 0:
            continue
        assert slices == F // 4 and G == -(-n_rows // 64) and PW == panels * W and 1 <= W <= 16 and 1 <= R <= 10
        assert (R - 1) * PW < G <= R * PW, (n_src, n_rows, F, list(plan))
        assert T % 16 == 0 and 16 <= T <= 10208 and T * NT >= n_src and (NT - 1) * T < n_src + 16 * NT
        ws = lib.dif_sliced_spmm_workspace_bytes(n_src, n_rows, F)
        if ws:
            S = ws // (slices * G * 64 * 16)
            assert S in (2, 4, 8) and ws == S * slices * G * 64 * 16 and NT % S == 0 and n_src + n_src // 16 >= S * n_rows
            assert panels * S * slices <= 256 or panels == 1
        elif n_src == n_rows:
            assert True                                     # a whole-graph product never splits


def test_argument_checks_reject_before_touching_the_device(lib):
    """Bad arguments return a negative DIF_E_* code and set dif_last_error; nothing is launched."""
    rc = lib.dif_simple_reduce_f32(None, 64, None, 64, None, 64, 10, 1, 64, 64, None, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.dif_last_error()
    rc = lib.dif_simple_reduce_f32(None, 64, None, 64, None, 64, 0, 1, 64, 64, None, None, 0, None)
    assert rc == -1 and b"positive" in lib.dif_last_error()
    rc = lib.dif_sigmoid_attn_f32(None, 8, None, 8, None, 8, 4, 4, 1, 8, 8, None, 8, None, 0, None)
    assert rc == -1
    rc = lib.dif_csr_build(None, 10, 0, None, 1, 0, 0, None, None, None, None, None, None, 0, None)
    assert rc == -1
    rc = lib.dif_csr_build(None, 2 ** 31, 10, None, 1, 0, 0, None, None, None, None, None, None, 0, None)
    assert rc == -4                                                                    # DIF_E_RANGE
    rc = lib.dif_gcn_spmm_f32(None, None, 1, None, None, 10, 5, None, 8, 0, 20, 8, None, 0, 1.0, 1.0, None, 0, None, 8, None)
    assert rc == -1 and b"row range" in lib.dif_last_error()
    rc = lib.dif_layer_tail_f32(None, 8, 4, 1, 8, None, 0, None, 0, 0.5, None, None, 1e-5, 0, None, 8, None)
    assert rc == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from difformer_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU / eager fallback"):
        _lib.load()


def test_tiny_model_entry_points_check_their_arguments(lib):
    """dif_tiny_*: size helpers are host arithmetic; the configuration is validated before any HIP call."""
    from difformer_amd import _lib
    def splits(n):                                  # csrc/tiny_common.h key_splits
        G = (n + 63) // 64
        return 1 if n <= 512 else max(1, min((256 + G - 1) // G, n // 64, 16))

    def tape(n, dp, L):                             # H, Z [L+1]; ATT, ATTLO [L]; two q/k/v sets; DEN; SUM; the splits' sums
        f = n * dp * (2 * (L + 1) + 2 * L + 6) + L * n + L * 96
        return f + (-f) % 4 + max(2 * splits(n) * n * (dp + 1), 4 * ((n + 63) // 64) * 96)

    def scratch(n, dp, L):
        one = n * dp * 14 + 3 * n
        g = n * dp * (5 * L + 15) + 4 * n
        return max(one, g + (-g) % 4 + max(2 * splits(n) * n * 3 * dp, 4 * ((n + 63) // 64) * 96))

    assert lib.dif_tiny_tape_floats(20, 4, 2) == tape(20, 4, 2)
    assert lib.dif_tiny_tape_floats(20, 5, 2) == tape(20, 8, 2)                                    # hidden 5..8: padded to 8
    assert lib.dif_tiny_tape_floats(1068, 4, 2) == tape(1068, 4, 2) and splits(1068) == 16 and splits(4096) == 4
    assert lib.dif_tiny_scratch_floats(1068, 4, 2) == scratch(1068, 4, 2)
    assert lib.dif_tiny_scratch_floats(20, 8, 1) == scratch(20, 8, 1)
    assert lib.dif_tiny_graph_workspace_bytes(100, 20) >= 4 * 100 * 4
    cfg = _lib.TinyCfg(n=5000, in_channels=4, hidden=4, out_channels=1, num_layers=2, kernel=0, use_bn=1, use_residual=1,
                       use_weight=0, use_graph=0, use_source=0, training=0, alpha=0.5, attn_scale=1.0, gcn_scale=1.0, dropout=0.0,
                       eps=1e-5, nnz=0)
    assert lib.dif_tiny_forward_f32(ctypes.byref(cfg), None, 4, None, None, None, None, None, None, None, None) == -2
    assert b"nodes" in lib.dif_last_error()
    cfg.n, cfg.hidden = 20, 9
    assert lib.dif_tiny_forward_f32(ctypes.byref(cfg), None, 4, None, None, None, None, None, None, None, None) == -2
    cfg.hidden = 4
    assert lib.dif_tiny_forward_f32(ctypes.byref(cfg), None, 4, None, None, None, None, None, None, None, None) == -1
    cfg.launch_plan = 3                             # 0 by size, 1 one workgroup, 2 grid
    assert lib.dif_tiny_forward_f32(ctypes.byref(cfg), None, 4, None, None, None, None, None, None, None, None) == -1
    assert b"launch_plan" in lib.dif_last_error()
    cfg.launch_plan = 0
    assert lib.dif_tiny_graph_build(None, None, 70000, 20, None, None, None, None, None, None, None, None, 0, None) == -2
    assert lib.dif_tiny_graph_build(None, None, 10, 20, None, None, None, None, None, None, None, None, 0, None) == -1
