"""State-dict key remapping between the reference's checkpoints and this package.

The reference saves ``module.state_dict()`` (train.py:1223) and loads the stage-1 MLLM checkpoint with
``mllm.load_state_dict(strict=True)`` (train.py:1137-1138).  With ``peft`` the Llama keys carry PEFT's
wrappers (``base_model.model.`` prefix, ``.base_layer`` on adapted linears, ``lora_A.default`` /
``lora_B.default`` adapters); the reference's own ``adjust_state_dict``
(ablation_study_without_lora.py:1071-1084) strips exactly those to reach the plain layout this package
uses (with adapters kept as ``...{q,v}_proj.lora_{A,B}.weight``).
"""
import re

_LLAMA = "llama_wrapper.llama_model."


def from_reference(state_dict, keep_lora=True):
    """PEFT- or plain-layout reference keys -> this package's keys (works for ``mllm.``-prefixed whole-model
    dicts and for bare MLLM dicts alike)."""
    out = {}
    for k, v in state_dict.items():
        nk = k.replace(_LLAMA + "base_model.model.", _LLAMA)
        nk = nk.replace(".base_layer.", ".")
        nk = re.sub(r"\.lora_([AB])\.[A-Za-z0-9_]+\.weight$", r".lora_\1.weight", nk)
        if ".lora_" in nk and not keep_lora:
            continue
        if "lora_dropout" in nk or "lora_embedding" in nk:
            continue
        out[nk] = v
    return out


def to_reference(state_dict, peft=True, adapter="default"):
    """This package's keys -> the layout a ``peft``-wrapped reference model expects (or the plain no-LoRA
    layout with adapters dropped when peft=False)."""
    adapted = set()
    for k in state_dict:
        m = re.match(r"(.*)\.lora_[AB]\.weight$", k)
        if m:
            adapted.add(m.group(1))
    out = {}
    for k, v in state_dict.items():
        is_lora = re.search(r"\.lora_([AB])\.weight$", k)
        if is_lora and not peft:
            continue
        nk = k
        if peft and _LLAMA in k:
            if is_lora:
                nk = re.sub(r"\.lora_([AB])\.weight$", rf".lora_\1.{adapter}.weight", k)
            else:
                base = k.rsplit(".", 1)[0]
                if base in adapted:
                    nk = base + ".base_layer." + k.rsplit(".", 1)[1]
            nk = nk.replace(_LLAMA, _LLAMA + "base_model.model.", 1)
        out[nk] = v
    return out
