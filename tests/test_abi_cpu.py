"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports
every symbol include/tcavt.h declares; the product path refuses to run without it; the host-side
mirror exposes the reference's state-dict key layout."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tcavt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tcavt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from tcavt_amd import capi

    declared = _declared_symbols()
    assert len(declared) >= 15
    handle = ctypes.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/tcavt.h but not exported"
    assert sorted(capi.EXPORTED_SYMBOLS) == declared, "capi.py binds a different symbol set than the header declares"
    assert capi.lib().tcavt_abi_version() == capi.ABI_VERSION == 4


def _struct_fields(name):
    text = open(os.path.join(ROOT, "include", "tcavt.h")).read()
    body = text[text.index("typedef struct %s {" % name):text.index("} %s;" % name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.split("{")[-1].strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?[A-Za-z0-9_]+(\s+const)?\s*\**(\s*const)?", "", decl)
        names += [p.strip().lstrip("*").strip() for p in decl.split(",") if p.strip()]
    return names


@pytest.mark.parametrize("cname,mirror", [("tcavt_gemm_args", "GemmArgs"), ("tcavt_llama_layer", "LlamaLayer"),
                                          ("tcavt_llama_stack_args", "LlamaStackArgs"),
                                          ("tcavt_llama_bwd_layer", "LlamaBwdLayer"),
                                          ("tcavt_llama_backward_args", "LlamaBackwardArgs"),
                                          ("tcavt_sample_params", "SampleParams"), ("tcavt_decode_args", "DecodeArgs"),
                                          ("tcavt_tlayer", "TLayer"), ("tcavt_tstack_args", "TStackArgs"),
                                          ("tcavt_cross_attn_args", "CrossAttnArgs"), ("tcavt_ltsf_args", "LtsfArgs"),
                                          ("tcavt_cross_attn_bwd_args", "CrossAttnBwdArgs"),
                                          ("tcavt_ltsf_bwd_args", "LtsfBwdArgs"),
                                          ("tcavt_tlayer_grads", "TLayerGrads"), ("tcavt_tstack_bwd_args", "TStackBwdArgs")])
def test_struct_mirrors_match_header_layout(cname, mirror, tmp_path):
    """Field order of each ctypes mirror follows the C struct, and -- compiled with the host C compiler against the real
    header -- so do sizeof and every field offset."""
    import subprocess

    from tcavt_amd import capi

    cls = getattr(capi, mirror)
    names = _struct_fields(cname)
    assert names == [f[0] for f in cls._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tcavt.h"\nint main(void) {\n'
                   + f'  printf("%zu\\n", sizeof({cname}));\n'
                   + "".join(f'  printf("%zu\\n", offsetof({cname}, {n}));\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(cls)
    assert out[1:] == [getattr(cls, n).offset for n in names]


def test_argument_errors_are_reported_not_crashed():
    """Contract violations return an error code + message before anything touches a device."""
    from tcavt_amd import capi

    args = capi.GemmArgs()
    rc = capi.lib().tcavt_gemm_bf16(ctypes.byref(args), None)
    assert rc == 1
    assert b"null" in capi.lib().tcavt_last_error()
    args.A = args.W = args.C = 64
    args.M, args.N, args.K = 8, 16, 60
    rc = capi.lib().tcavt_gemm_bf16(ctypes.byref(args), None)
    assert rc == 1 and b"multiple of 64" in capi.lib().tcavt_last_error()
    # round-3 stage entries: shapes outside the resident forms and missing pointers are refused, not launched
    assert capi.lib().tcavt_attn_bwd_resident_ok(256, 32, 8) == 1 and capi.lib().tcavt_attn_bwd_resident_ok(257, 32, 8) == 0
    assert capi.lib().tcavt_attn_bwd_resident_ok(256, 12, 4) == 0  # (16 % 3 != 0)
    rc = capi.lib().tcavt_attn_bwd_resident(64, 64, 64, 64, 64, 64, 64, 64, 64, 1, 300, 4, 1, 64, 0.125, capi.F16, None)
    assert rc == 1 and b"outside the resident form" in capi.lib().tcavt_last_error()
    bargs = capi.LlamaBackwardArgs()
    rc = capi.lib().tcavt_llama_stack_backward(ctypes.byref(bargs), None)
    assert rc == 1 and b"null" in capi.lib().tcavt_last_error()
    # the decode step's LoRA partial sums belong to the skinny form (M <= 32)
    args = capi.GemmArgs()
    args.A = args.W = args.C = args.lora_part = 64
    args.M, args.N, args.K = 64, 128, 256
    args.lda = args.ldw = 256
    args.ldc = 128
    rc = capi.lib().tcavt_gemm_bf16(ctypes.byref(args), None)
    assert rc == 1 and b"skinny form" in capi.lib().tcavt_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from tcavt_amd import capi

    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libtcavt_hip.so")
    with pytest.raises(capi.TcavtError, match="no CPU/PyTorch fallback"):
        capi.lib()


def test_cpu_tensors_are_rejected():
    from tcavt_amd import capi, ops

    a = torch.zeros(64, 64, dtype=torch.bfloat16)
    with pytest.raises(capi.TcavtError, match="must live on the GPU"):
        ops.gemm_bf16(a, a)


def test_state_dict_layout_matches_reference_keys():
    """Keys = the reference's MultiModalTrajectoryModel.state_dict() (fixtures were produced by
    load_state_dict(strict=True) of exactly these keys into the reference modules)."""
    from tcavt_amd import config, model
    from tcavt_amd.weights import make_weights

    for lora in (True, False):
        cfg = config.tiny(use_lora=lora)
        w = make_weights(cfg, 0)
        m = model.MultiModalTrajectoryModel.from_config(cfg)
        sd = m.state_dict()
        assert set(sd) == set(w)
        for k, v in w.items():
            assert tuple(sd[k].shape) == v.shape, k
        m.load_weights(w)
        k = "ltsf.decoder.cross_attn.in_proj_weight"
        assert torch.equal(m.state_dict()[k], torch.from_numpy(w[k]))
    assert m.mllm.llama_wrapper.llama_model.get_input_embeddings().weight.shape == (cfg.llama.vocab, cfg.llama.hidden)


def test_ctor_signature_mirrors_reference():
    import inspect

    from tcavt_amd import model

    sig = inspect.signature(model.MultiModalTrajectoryModel.__init__)
    ref = ["seq_len", "out_len", "individual", "feature_size", "d_model", "lane_polygon_d_model",
           "lane_polygon_nhead", "lane_polygon_layers", "max_polygon_points", "use_post_mlp", "post_mlp_hidden_dim",
           "base_model_name", "use_lora", "lora_r", "lora_alpha", "lora_dropout", "vision_dim", "q_hidden_size",
           "q_nhead", "q_enc_layers", "q_dec_layers", "q_num_query_tokens", "ltsf_nhead", "ltsf_dropout"]
    assert list(sig.parameters)[1:1 + len(ref)] == ref  # train.py:848-872
    fwd = list(inspect.signature(model.MultiModalTrajectoryModel.forward).parameters)[1:]
    assert fwd == ["x", "vision_embs", "context_str", "lane_polygon_batch", "lane_polygon_len", "y", "norm_stat",
                   "input_ids", "attention_mask", "labels"]  # train.py:914-924
