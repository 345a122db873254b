"""The oracle (numpy/C restatement) pinned against outputs of the reference itself.

Fixtures: tests/golden/*.npz, produced by tests/golden/make_golden.py which imports
`/root/reference/node classification/difformer.py` verbatim.  Tolerances: float64
restatement vs float64 reference run 1e-12 (same arithmetic, different summation
order); float32 vs float32 1e-5 norm-wise.
"""
import numpy as np
import pytest

from conftest import load_golden, rel_err, split_model_case
from oracle import difformer_oracle as orc

ATTN = load_golden("attn")
ATTNW = load_golden("attnw")
GCN = load_golden("gcn")
MODEL = load_golden("model")


@pytest.mark.parametrize("name", sorted(ATTN))
def test_full_attention_conv(name):
    c = ATTN[name]
    kern = str(c["kernel"])
    for suffix, dt, tol in (("f64", np.float64, 1e-12), ("f32", np.float32, 1e-5)):
        out = orc.full_attention_conv(c["q"].astype(dt), c["k"].astype(dt), c["v"].astype(dt), kern)
        assert out.dtype == dt and out.shape == c["out_" + suffix].shape
        assert rel_err(out, c["out_" + suffix]) < tol


@pytest.mark.parametrize("name", sorted(ATTNW))
def test_attention_weights(name):
    c = ATTNW[name]
    kern = str(c["kernel"])
    out, att = orc.full_attention_conv(c["q"].astype(np.float64), c["k"].astype(np.float64),
                                       c["v"].astype(np.float64), kern, output_attn=True)
    assert rel_err(out, c["out_f64"]) < 1e-12
    assert rel_err(att, c["attn_f64"]) < 1e-12


def test_simple_requires_equal_lengths():
    q = np.zeros((4, 1, 8), np.float32); k = np.zeros((5, 1, 8), np.float32)
    with pytest.raises(ValueError):
        orc.simple_attention(q, k, k)


@pytest.mark.parametrize("name", sorted(GCN))
def test_gcn_conv(name):
    c = GCN[name]
    w = c.get("edge_weight")
    for suffix, dt, tol in (("f64", np.float64, 1e-12), ("f32", np.float32, 1e-5)):
        out = orc.gcn_conv(c["x"].astype(dt), c["edge_index"], None if w is None else w.astype(dt))
        assert rel_err(out, c["out_" + suffix]) < tol


@pytest.mark.parametrize("name", sorted(GCN))
def test_gcn_conv_c_restatement(name):
    """oracle/gcn_conv_ref.c (used above 1e5 edges) agrees with the fixtures too."""
    lib = orc._load_clib()
    if not lib:
        pytest.skip("oracle/_build/liboracle_gcn.so not built (run __graft_entry__.build())")
    c = GCN[name]
    w = c.get("edge_weight")
    n, h, d = c["x"].shape
    e = c["edge_index"].shape[1]
    for suffix, dt, fn, tol in (("f64", np.float64, lib.oracle_gcn_conv_f64, 1e-12),
                                ("f32", np.float32, lib.oracle_gcn_conv_f32, 1e-5)):
        x = np.ascontiguousarray(c["x"].astype(dt)).reshape(n, h * d)
        ei = np.ascontiguousarray(c["edge_index"], dtype=np.int64)
        ww = None if w is None else np.ascontiguousarray(w.astype(dt))
        out = np.full_like(x, 7.0)
        for threads in (1, 3):
            rc = fn(x.ctypes.data, ei.ctypes.data, None if ww is None else ww.ctypes.data, n, e, h * d,
                    out.ctypes.data, threads)
            assert rc == 0
            assert rel_err(out.reshape(n, h, d), c["out_" + suffix]) < tol


@pytest.mark.parametrize("name", sorted(MODEL))
def test_model_forward(name):
    c = MODEL[name]
    cfg, sd = split_model_case(c)
    ei = c["edge_index"] if cfg["use_graph"] else None
    w = c.get("edge_weight")
    for suffix, dt, tol in (("f64", np.float64, 1e-11), ("f32", np.float32, 2e-5)):
        p = orc.cast_params(sd, dt)
        out, layers = orc.difformer_forward(p, c["x"].astype(dt), ei, None if w is None else w.astype(dt), cfg,
                                            return_layers=True)
        assert rel_err(out, c["out_" + suffix]) < tol
        conv0 = orc.difformer_conv(p, "convs.0.", layers[0], layers[0], ei,
                                   None if w is None else w.astype(dt), layers[0], cfg)
        assert rel_err(conv0, c["conv0_" + suffix]) < tol


def test_subgraph_restatement_small_case():
    """oracle.subgraph against a hand-checked case (torch_geometric.utils.subgraph semantics, main-batch.py:131)."""
    ei = np.array([[0, 1, 2, 3, 3, 4], [1, 2, 3, 4, 0, 4]])
    out, w = orc.subgraph(np.array([3, 0, 4]), ei, np.arange(6.0), relabel_nodes=True, num_nodes=5)
    assert out.tolist() == [[0, 0, 2], [2, 1, 2]] and w.tolist() == [3.0, 4.0, 5.0]
    out2, _ = orc.subgraph(np.array([3, 0, 4]), ei, None, relabel_nodes=False, num_nodes=5)
    assert out2.tolist() == [[3, 3, 4], [4, 0, 4]]


# ---- f4: DIFFormer_v2 (physical particle/difformer-v2.py), fixtures from tests/golden/make_golden_v2.py ----------
V2 = load_golden("v2")


@pytest.mark.parametrize("name", sorted(n for n in V2 if n.startswith("attn/")))
def test_v2_full_attention(name):
    c = V2[name]
    fn = orc.v2_simple_attention if str(c["kernel"]) == "simple" else orc.v2_sigmoid_attention
    for suffix, dt, tol in (("f64", np.float64, 1e-12), ("f32", np.float32, 1e-5)):
        out = fn(c["q"].astype(dt), c["k"].astype(dt), c["v"].astype(dt), c["n_nodes"])
        assert out.dtype == dt and out.shape == c["out_" + suffix].shape
        assert rel_err(out, c["out_" + suffix]) < tol


def test_v2_single_graph_is_a1():
    """One graph in the batch: difformer-v2.py:80-111 reduces to difformer.py:18-39."""
    rng = np.random.default_rng(0)
    q, k, v = (rng.standard_normal((50, 2, 16)) for _ in range(3))
    assert rel_err(orc.v2_simple_attention(q, k, v, [50]), orc.simple_attention(q, k, v)) < 1e-13


@pytest.mark.parametrize("name", sorted(n for n in V2 if n.startswith("model/")))
def test_v2_model_forward(name):
    c = V2[name]
    cfg, sd = split_model_case(c)
    ei = c["edge_index"] if cfg["use_graph"] else None
    for suffix, dt, tol in (("f64", np.float64, 1e-11), ("f32", np.float32, 2e-5)):
        out = orc.difformer_v2_forward(orc.cast_params(sd, dt), c["x"].astype(dt), ei, c["n_nodes"], cfg)
        assert rel_err(out, c["out_" + suffix]) < tol
