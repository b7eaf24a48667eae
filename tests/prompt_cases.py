"""Shared description of the prompt-assembly golden cases (tests/golden/prompt_tiny.npz, oracle/gen_golden.py)."""
import torch

CASES = {"cv_ns": (True, ["vivian", "ryan", "vivian"], ["chinese", "english", "auto"], None),
         "cv_st": (False, ["vivian", "ryan", "vivian"], ["chinese", "english", "auto"], None),
         "vd_st": (False, None, ["auto", "english"], None),
         "vc_st": (False, None, ["english", "auto", "chinese"], [True, False, True]),
         "vc_ns": (True, None, ["english", "auto", "chinese"], [True, False, True])}


def load_case(g, name):
    """-> dict(ids, ins, languages, speakers, non_streaming_mode, ref_ids, voice_clone_prompt)."""
    ns, spk, langs, icl = CASES[name]
    B = len(langs)
    ids = [torch.from_numpy(g[f"{name}_ids{i}"]) for i in range(B)]
    ins = [torch.from_numpy(g[f"{name}_ins{i}"]) if f"{name}_ins{i}" in g else None for i in range(B)]
    ref_ids = vcp = None
    if icl is not None:
        ref_ids = [torch.from_numpy(g[f"{name}_refids{i}"]) for i in range(B)]
        vcp = dict(ref_code=[torch.from_numpy(g[f"{name}_refcode{i}"]) if f"{name}_refcode{i}" in g else None for i in range(B)],
                   ref_spk_embedding=[torch.from_numpy(g[f"{name}_spk{i}"]) for i in range(B)],
                   x_vector_only_mode=[not x for x in icl], icl_mode=list(icl))
    return dict(ids=ids, ins=ins, languages=langs, speakers=spk, non_streaming_mode=ns, ref_ids=ref_ids, voice_clone_prompt=vcp)
