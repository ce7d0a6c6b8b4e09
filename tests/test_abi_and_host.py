"""CPU: the C-ABI library loads and exports every declared symbol; host-side module logic
(constructor surface, state-dict layout, flat parameter buffer, loud failure without a GPU)."""
import inspect
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(REPO, "include", "toad_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(toad_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    from toad_amd import _lib
    lib = _lib.load()                       # dlopen only: no kernel is launched on this box
    names = header_functions()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/toad_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert lib.toad_abi_version() == _lib.ABI_VERSION


def test_argument_errors_are_reported_without_a_gpu():
    from toad_amd import _lib
    lib = _lib.load()
    rc = lib.toad_linear_act_fwd_f32(None, None, None, None, 4, 4, 4, 0, 0.0, 0, None, None, None, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.toad_last_error()
    # ABI v7: whole-slide entry points and the abs-max plumbing validate before touching the device
    import ctypes
    one = ctypes.c_void_p(1 << 21)
    err = lambda: lib.toad_last_error().decode()
    assert lib.toad_amax_floats(1) == 1 and lib.toad_amax_floats(256) == 1 and lib.toad_amax_floats(257) == 2 and lib.toad_amax_floats(100000) == 391
    assert lib.toad_absmax_rows256_f32(one, 10, 6, one, None) == -1 and "bad argument" in err()
    assert lib.toad_linear_h2_ok(100000, 512, 1024) == 1               # the headline shape runs on the fp16 two-piece kernels, no switch exists
    assert lib.toad_linear_h2_ok(100000, 512, 1000) == 0 and lib.toad_linear_h2_ok(2_000_000, 512, 1024) == 0
    assert lib.toad_mil_arena_bytes(1000, 18, 384) > 1000 * (512 + 512 + 768 + 2) * 4
    assert lib.toad_mil_arena_bytes(1000, 18, 100) == 0 and lib.toad_mil_scratch_bytes(0, 18, 384) == 0
    assert lib.toad_relu_bits_bytes(100000, 512) == 391 * 2 * 8192 and lib.toad_relu_bits_bytes(1, 4) == 8192
    offs = (ctypes.c_int64 * 18)()
    assert lib.toad_mil_arena_layout(100000, 18, 384, offs) == 0
    o = list(offs)
    assert o[0] == 0 and all(b > a for a, b in zip(o, o[1:])) and all(x % (1 << 21) == 0 for x in o[:4]) and o[15] - o[14] == 391 * 4
    assert lib.toad_mil_step_ws_bytes(100000, 18, 384) >= lib.toad_mil_arena_bytes(100000, 18, 384) + lib.toad_mil_scratch_bytes(100000, 18, 384)
    p12 = (ctypes.c_void_p * 12)(*([1 << 21] * 12))
    assert lib.toad_mil_fwd_f32(p12, one, one, 1000, 18, 100, 0.0, 0, None, 0, one, 1 << 40, one, 1 << 40, None) == -2 and "unsupported shape" in err()
    assert lib.toad_mil_fwd_f32(p12, one, one, 1000, 18, 384, 0.0, 0, None, 0, one, 16, one, 1 << 40, None) == -3 and "arena too small" in err()
    assert lib.toad_mil_fwd_f32(p12, one, one, 1000, 18, 384, 0.0, 0, None, 0, one, 1 << 40, one, 16, None) == -3 and "scratch too small" in err()
    p12[3] = None
    assert lib.toad_mil_fwd_f32(p12, one, one, 1000, 18, 384, 0.0, 0, None, 0, one, 1 << 40, one, 1 << 40, None) == -1 and "slot 3" in err()
    assert lib.toad_mil_bwd_f32(p12, p12, 0.0, one, 1000, 18, 384, 0.0, 0, one, 1 << 40, None, one, None, None, None, None, one, 1 << 40, None) == -1
    assert lib.toad_sgd_step_f32(one, one, None, 1024, 0.1, 0.9, 0.0, 1, None) == -1       # momentum without a buffer
    assert lib.toad_heads_ce_fused_f32(*([None] * 8), 0.75, 0.25, *([None] * 15), 0.0, 512, 18, None) == -1
    assert lib.toad_gated_pool_ws_bytes(1000, 512, 384, 2) > 0
    assert lib.toad_gated_pool_ws_bytes(1000, 500, 384, 2) == 0      # unsupported shape
    assert lib.toad_linear_wgrad_ws_bytes(100000, 512, 1024) >= 512 * 1024 * 4


def test_extractor_entry_points_validate_arguments_without_a_gpu():
    """Every new conv / extractor entry point rejects bad arguments before touching the device, with a message."""
    import ctypes
    from toad_amd import _lib
    lib = _lib.load()
    one = ctypes.c_void_p(16)                  # a non-null, 16-byte aligned fake pointer: argument checks come first
    err = lambda: lib.toad_last_error().decode()
    assert lib.toad_conv_nhwc_f32(None, None, None, None, None, 1, 8, 8, 64, 3, 3, 1, 1, 64, 1, None, 0, None) == -1 and "null pointer" in err()
    assert lib.toad_conv_nhwc_f32(one, one, None, None, one, 1, 8, 8, 48, 3, 3, 1, 1, 64, 1, one, 1 << 30, None) == -2 and "multiple of 32" in err()
    assert lib.toad_conv_nhwc_f32(one, one, None, None, one, 1, 8, 8, 64, 3, 3, 1, 1, 1024, 1, one, 1 << 30, None) == -2 and "Cout <= 512" in err()
    assert lib.toad_conv_nhwc_f32(one, one, None, None, one, 1, 2, 2, 64, 5, 5, 1, 0, 64, 1, one, 1 << 30, None) == -2 and "empty output" in err()
    assert lib.toad_conv_nhwc_f32(one, one, None, None, one, 1, 8, 8, 64, 3, 3, 1, 1, 64, 7, one, 1 << 30, None) == -1 and "bad act" in err()
    assert lib.toad_conv_nhwc_f32(one, one, None, None, one, 1, 8, 8, 64, 3, 3, 1, 1, 64, 1, one, 16, None) == -3 and "workspace too small" in err()
    assert lib.toad_im2col_nhwc_f32(one, one, 1, 8, 8, 6, 3, 3, 1, 1, None) == -2 and "multiple of 4" in err()
    assert lib.toad_im2col_nhwc_f32(ctypes.c_void_p(4), one, 1, 8, 8, 8, 3, 3, 1, 1, None) == -4 and "aligned" in err()
    assert lib.toad_maxpool3x3s2_nhwc_f32(one, None, 1, 8, 8, 64, None) == -1
    assert lib.toad_avgpool_nhwc_f32(one, one, 70000, 4, 64, None) == -2
    assert lib.toad_stem_s2d_nchw_f32(one, one, 0, 8, 8, None) == -2
    assert lib.toad_stem_conv_s2d_f32(one, one, None, one, 1, 4, 4, 1, one, 16, None) == -3
    assert lib.toad_linear_act_res_fwd_f32(one, one, None, ctypes.c_void_p(20), one, 8, 32, 64, 1, None, 0, None) == -2 and "residual" in err()
    assert lib.toad_resnet50_trunc_ws_bytes(0, 256, 256) == 0 and lib.toad_resnet50_trunc_ws_bytes(4, 256, 256) > 4 * 20e6
    w = (ctypes.c_void_p * 43)(*([16] * 43))
    assert lib.toad_resnet50_trunc_fwd_f32(one, w, w, one, 2, 64, 64, one, 1024, None) == -3 and "workspace too small" in err()
    w[7] = None
    big = lib.toad_resnet50_trunc_ws_bytes(2, 64, 64)
    assert lib.toad_resnet50_trunc_fwd_f32(one, w, w, one, 2, 64, 64, one, big, None) == -1 and "slot 7" in err()


def test_constructor_and_state_dict_match_reference(golden):
    from toad_amd import TOAD_fc_mtl_concat
    def plain(fn):      # "(self, gate=True, ...)" without annotations
        ps = inspect.signature(fn).parameters.values()
        return "(" + ", ".join(p.name if p.default is p.empty else f"{p.name}={p.default!r}" for p in ps) + ")"
    assert plain(TOAD_fc_mtl_concat.__init__) == str(golden["api/init_sig"])
    assert plain(TOAD_fc_mtl_concat.forward) == str(golden["api/forward_sig"])
    rows = str(golden["api/state_dict"]).split("\n")
    for c, dr in ((18, False), (2, True)):
        m = TOAD_fc_mtl_concat(dropout=dr, n_classes=c)
        mine = [f"{c}|{int(dr)}|{k}|{','.join(map(str, v.shape))}" for k, v in m.state_dict().items()]
        assert mine == [r for r in rows if r.startswith(f"{c}|{int(dr)}|")]
    with pytest.raises(NameError):
        TOAD_fc_mtl_concat(gate=False)      # reference: model_toad.py:68 references an undefined Attn_Net


def test_initialisation_is_xavier_normal_zero_bias():
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(1)
    m = TOAD_fc_mtl_concat(n_classes=18)
    for k, v in m.state_dict().items():
        if k.endswith("bias"):
            assert float(v.abs().max()) == 0.0
        else:
            std = (2.0 / (v.shape[0] + v.shape[1])) ** 0.5
            assert abs(float(v.std()) - std) < 0.15 * std, k


def test_flat_parameter_buffer_roundtrip():
    from toad_amd import TOAD_fc_mtl_concat
    m = TOAD_fc_mtl_concat(n_classes=18)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    flat = m.flatten_parameters()
    assert m._is_flat()
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k])
    a = m.attention_net[4].attention_a[0].weight
    b = m.attention_net[4].attention_b[0].weight
    assert b.data_ptr() == a.data_ptr() + 4 * a.numel()          # [Wa;Wb] is one zero-copy view
    assert torch.equal(m._views["wab"], torch.cat([a, b], 0))
    assert all(p.data_ptr() % 256 == flat.data_ptr() % 256 or k in ("wb", "bb") for k, p in m._slot_params().items())
    m.load_state_dict(sd)                                          # in-place copy keeps the views
    assert m._is_flat()
    m.double().float()                                             # _apply re-homes parameters ...
    assert not m._is_flat()
    m.flat_parameters()                                            # ... and the module re-flattens on demand
    assert m._is_flat()
    offs, total = m.flat_offsets()
    assert total == flat.numel() and sum(n for _, n in offs.values()) == 1192490


def test_no_cpu_fallback():
    from toad_amd import TOAD_fc_mtl_concat, Attn_Net_Gated
    m = TOAD_fc_mtl_concat(n_classes=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(8, 1024), torch.zeros(1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Attn_Net_Gated()(torch.zeros(8, 1024))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device"):
            m.relocate()


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "toad_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_adjacent_bags_are_their_own_concatenation():
    """ops._adjacent_rows: consecutive row ranges of one allocation go to the ragged multi-slide call as a VIEW (no torch.cat copy); anything
    else (a gap, another allocation, another dtype, a permuted order) is not adjacent and gets concatenated as before."""
    import torch
    from toad_amd import ops
    pool = torch.arange(10 * 8, dtype=torch.float32).reshape(10, 8)
    a, b, c = pool[0:3], pool[3:4], pool[4:10]
    v = ops._adjacent_rows([a, b, c])
    assert v is not None and v.shape == (10, 8) and v.data_ptr() == pool.data_ptr() and torch.equal(v, pool)
    assert ops._adjacent_rows([pool[2:5], pool[5:9]]).data_ptr() == pool[2:].data_ptr()
    assert ops._adjacent_rows([a, c]) is None                               # a gap
    assert ops._adjacent_rows([b, a]) is None                               # wrong order
    assert ops._adjacent_rows([a, b.clone()]) is None                       # another allocation
    assert ops._adjacent_rows([pool.half()[0:3], pool.half()[3:4]]) is None  # not fp32
    assert ops._adjacent_rows([a, pool[3:4, :4]]) is None                   # another width / non-contiguous



def test_cached_weight_table_follows_the_module_tree():
    """TOAD_fc_mtl_concat._weights() caches the slot -> tensor table it hands to the library (walking the module tree cost 75 us per forward,
    a tenth of the reference loop's host time on a 10k-patch bag) and re-validates it per call by identity. Every way a caller can change what
    the model's parameters ARE must be picked up on the next call: a replaced head, a replaced layer inside attention_net, parameter storage
    moved by hand, a load_state_dict (in place: same objects, new values)."""
    import torch
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(0)
    m = TOAD_fc_mtl_concat(n_classes=18)
    w0 = m._weights()
    w1 = m._weights()
    assert all(w0[k] is w1[k] for k in ("w1", "wcls", "wc")) and w0["wab"].data_ptr() == w1["wab"].data_ptr()
    assert m._is_flat() and w0["wab"].shape == (768, 512) and w0["wab"].data_ptr() == w0["wa"].data_ptr()
    # a replaced head
    m.classifier = torch.nn.Linear(513, 18)
    w2 = m._weights()
    assert w2["wcls"] is m.classifier.weight and m._is_flat() and w2["wcls"].data_ptr() != w0["wcls"].data_ptr()
    # a replaced layer inside the Sequential
    m.attention_net[0] = torch.nn.Linear(1024, 512)
    w3 = m._weights()
    assert w3["w1"] is m.attention_net[0].weight and m._is_flat()
    # storage moved by hand (what .to() / a cast does): re-flattened, values kept
    keep = m.attention_net[0].bias.detach().clone()
    m.attention_net[0].bias.data = m.attention_net[0].bias.data.clone()
    w4 = m._weights()
    assert m._is_flat() and torch.equal(w4["b1"].detach(), keep)
    # in-place load: same objects, the table stays valid and shows the new values
    sd = {k: torch.full_like(v, 0.5) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    w5 = m._weights()
    assert w5["w2"] is w4["w2"] and float(w5["w2"].detach().mean()) == 0.5 and float(w5["wab"].mean()) == 0.5
