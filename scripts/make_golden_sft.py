"""Generate tests/golden/sft_conversations.json by executing the REFERENCE's own ``prepare_dataset`` (open_r1/sft.py:84-143,
extracted with ast -- the module itself imports trl / requests / accelerate, absent here) on crafted dataset rows, plus
its label rule applied to a toy id sequence.  Data only: inputs and the reference's outputs.

    python scripts/make_golden_sft.py
"""
import ast
import json
import os
from typing import Any, Dict, List

REF = "/root/reference/SpaceR-SG-RLVR/src/r1-v/src/open_r1/sft.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sft_conversations.json")

ROWS = [
    dict(problem="Which object is closest to the door?", options=["A. chair", "B. table", "C. sofa"], solution="<answer>B</answer>",
         path="/data/v1.mp4", data_type="video", problem_type="multiple choice"),
    dict(problem="How many chairs are in the room?", options=[], solution="<think>count</think><answer>4</answer>",
         path="/data/v2.mp4", data_type="video", problem_type="numerical"),
    dict(problem="What does the sign say?", options=[], solution="<answer>EXIT</answer>", path="/data/i1.jpg", data_type="image",
         problem_type="OCR"),
    dict(problem="Describe the scene.", options=[], solution="<answer>a kitchen</answer>", path="/data/v3.mp4", data_type="video",
         problem_type="free-form"),
    dict(problem="What is the distance between the sofa and the tv in meters?", options=[], solution="<answer>2.5</answer>",
         path="/data/v4.mp4", data_type="video", problem_type="regression"),
]


def main():
    tree = ast.parse(open(REF).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "prepare_dataset")
    ns = {"Dict": Dict, "List": List, "Any": Any}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
    out = [dict(row=r, messages=ns["prepare_dataset"](dict(r))["messages"]) for r in ROWS]
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, len(out), "rows")


if __name__ == "__main__":
    main()
