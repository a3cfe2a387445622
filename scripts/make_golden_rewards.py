"""Generate tests/golden/reward_table.json and tests/golden/vision_tables.json by importing the REFERENCE's own
pure-Python reward / map / frame-sampling code from /root/reference in this container (it cannot travel to the
GPU box; only these input/expected-output vectors do).  Missing third-party imports of the reference modules
(trl, trainer, nltk, rouge_score, jsonlines, torchvision, ...) are stubbed in sys.modules: none of them is
executed by the functions recorded here.

    python scripts/make_golden_rewards.py
"""
import importlib.util
import itertools
import json
import os
import sys
import types

REF = "/root/reference/SpaceR-SG-RLVR/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _stub(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference():
    class _Any:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return self
        def __getattr__(self, k): return _Any()
    for n in ("trl", "jsonlines", "nltk", "nltk.translate", "nltk.translate.bleu_score", "rouge_score", "trainer"):
        _stub(n)
    sys.modules["trl"].__dict__.update(GRPOConfig=_Any, GRPOTrainer=_Any, ModelConfig=_Any, ScriptArguments=object,
                                       TrlParser=_Any, get_peft_config=_Any)
    sys.modules["nltk.translate.bleu_score"].__dict__.update(sentence_bleu=_Any(), SmoothingFunction=_Any)
    sys.modules["rouge_score"].__dict__.update(rouge_scorer=_Any())
    sys.modules["trainer"].__dict__.update(SGRLVRTrainer=_Any, Qwen2VLGRPOTrainer=_Any, Qwen2VLGRPOVLLMTrainerModified=_Any)
    sys.path.insert(0, os.path.join(REF, "r1-v", "src", "open_r1"))

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    em = load(os.path.join(REF, "r1-v", "src", "open_r1", "extract_map.py"), "extract_map")
    sg = load(os.path.join(REF, "r1-v", "src", "open_r1", "SG-RLVR.py"), "sg_rlvr_ref")
    # torchvision / decord are stubbed only now: transformers (imported by SG-RLVR.py) probes torchvision at import
    for n in ("torchvision", "torchvision.io", "torchvision.transforms", "decord"):
        _stub(n)
    sys.modules["torchvision"].__dict__.update(io=sys.modules["torchvision.io"], transforms=sys.modules["torchvision.transforms"])
    sys.modules["torchvision.transforms"].__dict__.update(InterpolationMode=_Any(), functional=_Any())
    vp = load(os.path.join(REF, "qwen-vl-utils", "src", "qwen_vl_utils", "vision_process.py"), "vision_process_ref")
    return em, sg, vp


COG = {"vid_a": {"chair": [[1, 2], [5, 5]], "table": [[7, 3]], "tv": [[9, 9]]},
       "vid_b": {"sofa": [[2, 2]], "coffee table": [[4, 4], [6, 1]], "table": [[0, 9]], "door": [[8, 0]]}}

MAP_TEXTS = [
    '{"chair": [[1, 3], [5, 5]], "table": [[7, 3]], "tv": [[9, 8]]}',
    "{'chair': [[1,2]], 'table': [[7,3]], 'tv': [[0,0]]}",
    '{"Chair": [[1, 2], [5, 5], [9, 9]], "TABLE": [7, 3]}',
    'str{ {"chair": [[1, 2]], "tv": [["9", "9"]]} }',
    '{"chair": [(1, 2), (5.7, 5.2)], "lamp": [[1, 1]], "tv": "9, 9"}',
    '{"<chair>": [[1,2]], "\'table\'": [[7, 3]], "tv:": [[9, 9]]}',
    '{"chair": [[[1, 2]], [5, 5]], "table": [[7]], "tv": [[9, 9, 9]]}',
    "chair: [1, 2], [5, 5]; table: (7, 3); tv at 9,9",
    "The chair is at [1,3] and another Chair at [4,5]. table -> 7 3. TV: <9, 8>",
    "chairs everywhere 1 2, armchair 3 4, tv 9",
    '{"chair": [[1, 2], [5, 5]], "table": [[7, 3]], "tv": [[9, 9]]',
    "{chair: [[1,2]], table: [[7,3]]}",
    "sofa (2,2) coffee table (4,4) (6,1) table (0,9) door 8 0",
    "coffee table: [[4, 4]], table: [[0, 9]], sofa: [[2.9, 2.1]], door: [[-1, 0]]",
    '{"sofa": [[2, 2]], "coffee table": [[4, 4], [6, 1]], "table": [[0, 9]], "door": [[8, 0]]}',
    "", "no objects here", "{}", "[]", '{"chair": []}', "{'chair': [[1, 'x']]}",
]

ANSWERS_MC = ["A", " A ", "B", "a", "", "A.", "(A)"]
ANSWERS_NUM = ["3", "3.2", "about 3.2 m", "three", "a dozen", "twenty five", "an apple", "none", "-4", "1,200", "0", "2.999"]
ANSWERS_TXT = ["the quick brown fox", "the quick fox", "", "a completely different sentence here", "The quick brown fox"]


def build_cases():
    cases = []

    def add(kind, content, sol, path):
        cases.append({"problem_type": kind, "content": content, "solution": sol, "path": path})
    think = "<think>hmm</think>"
    for vid, cog in COG.items():
        path = f"/data/{vid}.mp4"
        for ans in ANSWERS_MC:
            add("multiple choice", f"{think}<answer>{ans}</answer>", "<answer>A</answer>", path)
        for mt, ans in itertools.product(MAP_TEXTS, ("A", "B")):
            add("multiple choice", f"<think>x</think>\n<map>{mt}</map>\n<answer>{ans}</answer>", "<answer> A </answer>", path)
        for ans, gt in itertools.product(ANSWERS_NUM, ("3", "3.0", "0", "25", "abc", "-4")):
            add("numerical", f"{think}<answer>{ans}</answer>", f"<answer>{gt}</answer>", path)
        for mt in MAP_TEXTS[:8]:
            add("numerical", f"<think>y</think><map>{mt}</map><answer>3.1</answer>", "<answer>3</answer>", path)
            add("numerical", f"<think>y</think><map>{mt}</map><answer>9</answer>", "<answer>3</answer>", path)
    for ans, gt in itertools.product(ANSWERS_TXT, ANSWERS_TXT[:2] + [""]):
        add("OCR", f"<think>t</think><answer>{ans}</answer>", f"<answer>{gt}</answer>", "/d/vid_a.mp4")
    for ans, gt in itertools.product(["3.5", "1,000", "abc", "-2", "0", "3.5 m"], ["3.5", "1000", "0", "-2.5", "x"]):
        add("regression", f"<answer>{ans}</answer>", f"<answer>{gt}</answer>", "/d/vid_a.mp4")
    add("unknown type", "<answer>A</answer>", "<answer>A</answer>", "/d/vid_a.mp4")
    add("multiple choice", "no tags at all", "<answer>A</answer>", "/d/vid_a.mp4")
    add("multiple choice", "<map>{}</map><answer>A</answer>", "<answer>A</answer>", "/d/unknown_video.mp4")
    return cases


FORMAT_CASES = [
    "<think>a</think><answer>b</answer>", "<think>a</think>\n\n<answer>b</answer>", "<think>a</think> <map>{}</map> <answer>b</answer>",
    " <think>a</think><answer>b</answer>", "<think>a</think><answer>b</answer> ", "<think></think><answer></answer>",
    "<think>a\nb</think>\n<answer>c\nd</answer>", "<answer>b</answer>", "<think>a</think>", "", "<think>a</think><answer>b</answer><answer>c</answer>",
]


def main():
    em, sg, vp = load_reference()
    sg.MAP_DATA = {k: {"cognitive_map": v, "object_list": list(v)} for k, v in COG.items()}
    cases = build_cases()
    for c in cases:
        r = sg.accuracy_reward([[{"role": "assistant", "content": c["content"]}]], [c["solution"]], [c["path"]],
                               problem_type=[c["problem_type"]])
        c["reward"] = float(r[0])
    maps = []
    for vid, cog in COG.items():
        for mt in MAP_TEXTS:
            parsed = em.extract_map_data(mt, list(cog))
            try:
                score = em.calculate_prediction_score(parsed, cog, 10)
            except Exception as e:           # noqa: BLE001
                score = f"raises:{type(e).__name__}"
            maps.append({"video": vid, "text": mt, "parsed": parsed, "score": score})
    extra_scores = []
    for resp, sol in [({}, {}), ({"a": []}, {"a": []}), ({"a": [[1, 1]]}, {}), ({}, {"a": [[1, 1]]}),
                      ({"a": [[0, 0], [9, 9], [5, 5]]}, {"a": [[9, 9], [0, 1]]}), ({"a": [[1, 3]]}, {"a": [[1, 2], [5, 5]]}),
                      ({"a": [[0, 0]], "b": [[3, 3]]}, {"a": [[10, 10]], "b": [[3, 4], [3, 3]]})]:
        try:
            v = em.calculate_prediction_score(resp, sol, 10)
        except Exception as e:               # noqa: BLE001
            v = f"raises:{type(e).__name__}"
        extra_scores.append({"response": resp, "solution": sol, "score": v})
    fmt = [{"content": c, "reward": sg.format_reward([[{"content": c}]])[0]} for c in FORMAT_CASES]
    with open(os.path.join(OUT, "reward_table.json"), "w") as f:
        json.dump({"cognitive_maps": COG, "accuracy": cases, "maps": maps, "scores": extra_scores, "format": fmt}, f, indent=0)
    print(f"reward_table.json: {len(cases)} accuracy rows, {len(maps)} map rows, {len(fmt)} format rows")

    # ---- frame-sampling tables
    resize = []
    for h, w in [(480, 640), (720, 1280), (448, 448), (1080, 1920), (360, 640), (240, 320), (100, 100), (2160, 3840), (56, 5000)]:
        for mn, mx in [(vp.VIDEO_MIN_PIXELS, int(vp.VIDEO_MIN_PIXELS * 1.05)), (vp.MIN_PIXELS, vp.MAX_PIXELS), (3136, 12845056)]:
            resize.append({"h": h, "w": w, "min_pixels": mn, "max_pixels": mx, "out": list(vp.smart_resize(h, w, 28, mn, mx))})
    nfr = []
    for total, fps in [(300, 30), (120, 24), (50, 25), (9, 30), (4, 1), (1000, 29.97), (33, 15), (2, 30), (17, 8)]:
        for ele in ({}, {"fps": 1.0}, {"nframes": 7}, {"fps": 4.0, "max_frames": 12}, {"min_frames": 6}):
            try:
                n = vp.smart_nframes(dict(ele), total, fps)
                idx = __import__("torch").linspace(0, total - 1, n).round().long().tolist()
                out = {"nframes": n, "idx": idx}
            except Exception as e:           # noqa: BLE001
                out = {"raises": type(e).__name__}
            nfr.append({"total": total, "fps": fps, "ele": ele, **out})
    budget = []
    for n in (4, 8, 16, 32, 64, 768):
        for ele in ({}, {"max_pixels": 50176}, {"total_pixels": 20480 * 28 * 28}, {"min_pixels": 64 * 28 * 28}):
            mn = ele.get("min_pixels", vp.VIDEO_MIN_PIXELS)
            tot = ele.get("total_pixels", vp.VIDEO_TOTAL_PIXELS)
            mx = max(min(vp.VIDEO_MAX_PIXELS, tot / n * vp.FRAME_FACTOR), int(mn * 1.05))
            mx = min(ele.get("max_pixels", mx), mx)
            budget.append({"nframes": n, "ele": ele, "max_pixels": mx})
    # ---- image inputs: the reference's own fetch_image (:99-142) on seeded images -> output size + sha256 of the RGB bytes
    import hashlib
    import numpy as np
    from PIL import Image
    images = []
    for seed, (h, w), mode, ele in [(0, (100, 150), "RGB", {}), (1, (480, 640), "RGB", {}), (2, (37, 901), "RGB", {}),
                                    (3, (64, 64), "RGBA", {}), (4, (200, 300), "L", {}), (5, (300, 200), "RGB", {"max_pixels": 50176}),
                                    (6, (90, 120), "RGB", {"resized_height": 280, "resized_width": 420}),
                                    (7, (1200, 1600), "RGB", {}), (8, (20, 30), "RGB", {"min_pixels": 3136})]:
        ch = {"RGB": 3, "RGBA": 4, "L": 1}[mode]
        arr = np.random.RandomState(seed).randint(0, 256, (h, w, ch), dtype=np.uint8)
        img = Image.fromarray(arr.squeeze(), mode)
        out = vp.fetch_image({"image": img, **ele})
        images.append({"seed": seed, "h": h, "w": w, "mode": mode, "ele": ele, "size": list(out.size),
                       "sha256": hashlib.sha256(np.asarray(out).tobytes()).hexdigest()})
    with open(os.path.join(OUT, "vision_tables.json"), "w") as f:
        json.dump({"constants": {k: getattr(vp, k) for k in ("VIDEO_MIN_PIXELS", "VIDEO_MAX_PIXELS", "FPS", "FPS_MIN_FRAMES",
                                                              "FPS_MAX_FRAMES", "FRAME_FACTOR", "VIDEO_TOTAL_PIXELS")},
                   "smart_resize": resize, "smart_nframes": nfr, "budget": budget, "fetch_image": images}, f, indent=0)
    print(f"vision_tables.json: {len(resize)} resize rows, {len(nfr)} nframes rows, {len(budget)} budget rows, {len(images)} image rows")


if __name__ == "__main__":
    main()
