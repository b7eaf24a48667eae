#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Writes tests/golden/api_surface.json: the method names, parameter names and literal defaults
of the reference's public wrappers (qwen_tts/inference/qwen3_tts_model.py: Qwen3TTSModel, VoiceClonePromptItem;
qwen_tts/inference/qwen3_tts_tokenizer.py: Qwen3TTSTokenizer), read with `ast` from /root/reference -- nothing is
imported or executed.  tests/test_host_logic.py::test_api_surface_matches_reference compares the mirrored classes with
this file, so the drop-in boundary is pinned to the reference rather than to memory.

    python oracle/gen_api_surface.py          # needs /root/reference (this container only)
"""
import ast
import json
import os

REF = os.environ.get("QTTS_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = {
    "qwen_tts/inference/qwen3_tts_model.py": ["Qwen3TTSModel", "VoiceClonePromptItem"],
    "qwen_tts/inference/qwen3_tts_tokenizer.py": ["Qwen3TTSTokenizer"],
}


def _default(node):
    try:
        return repr(ast.literal_eval(node))
    except Exception:
        return ast.unparse(node)


def surface(path, classes):
    tree = ast.parse(open(path).read())
    out = {}
    for n in tree.body:
        if not (isinstance(n, ast.ClassDef) and n.name in classes):
            continue
        fields = [s.target.id for s in n.body if isinstance(s, ast.AnnAssign) and isinstance(s.target, ast.Name)]
        methods = {}
        for f in n.body:
            if not isinstance(f, ast.FunctionDef):
                continue
            a = f.args
            pos = a.posonlyargs + a.args
            defaults = [None] * (len(pos) - len(a.defaults)) + [_default(d) for d in a.defaults]
            params = [[p.arg, d] for p, d in zip(pos, defaults)]
            if a.vararg:
                params.append(["*" + a.vararg.arg, None])
            params += [[p.arg, None if d is None else _default(d)] for p, d in zip(a.kwonlyargs, a.kw_defaults)]
            if a.kwarg:
                params.append(["**" + a.kwarg.arg, None])
            deco = [ast.unparse(d) for d in f.decorator_list if ast.unparse(d) in ("classmethod", "staticmethod", "property")]
            methods[f.name] = {"params": params, "kind": deco[0] if deco else "method", "line": f.lineno}
        out[n.name] = {"fields": fields, "methods": methods}
    return out


def main():
    doc = {"source": "QwenLM/Qwen3-TTS (ast of the files below; see oracle/gen_api_surface.py)", "classes": {}}
    for rel, classes in FILES.items():
        for name, body in surface(os.path.join(REF, rel), classes).items():
            body["file"] = rel
            doc["classes"][name] = body
    dst = os.path.join(ROOT, "tests", "golden", "api_surface.json")
    with open(dst, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("wrote", dst, {k: len(v["methods"]) for k, v in doc["classes"].items()})


if __name__ == "__main__":
    main()
