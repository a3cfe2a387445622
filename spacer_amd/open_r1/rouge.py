"""ROUGE-1 / ROUGE-2 / ROUGE-L F-measures for the ``free-form`` problem type (reference: SG-RLVR.py:158-162,208-210, which calls
``rouge_score.rouge_scorer.RougeScorer(['rouge1', 'rouge2', 'rougeL'], use_stemmer=True).score(reference, hypothesis)``).

``rouge_score`` (and the nltk stemmer under it) is a third-party package that ships neither with the reference nor in this image.
When it is importable it is used as is -- exact parity by construction.  Otherwise this file restates its published algorithm:
  tokenise      lower-case, every run of characters outside [a-z0-9] becomes a separator, Porter-stem tokens longer than 3
  rouge-n       clipped n-gram overlap; precision = overlap / #prediction n-grams, recall = overlap / #reference n-grams
  rouge-L       longest common subsequence of the two token lists; precision = lcs / #prediction, recall = lcs / #reference
  F-measure     2PR / (P + R), 0 when P + R = 0
The stemmer is M. Porter's 1980 algorithm with the nltk-extension irregular-form table.  PARITY WITH THE PACKAGE IS UNPINNED (it
cannot be run here); the restatement is pinned by hand-computed cases in tests/test_rewards_golden.py.
"""
from __future__ import annotations

import re
from collections import Counter
from typing import List

_IRREGULAR = {"sky": "sky", "skies": "sky", "dying": "die", "lying": "lie", "tying": "tie", "news": "news", "inning": "inning",
              "innings": "inning", "outing": "outing", "outings": "outing", "canning": "canning", "cannings": "canning",
              "howe": "howe", "proceed": "proceed", "exceed": "exceed", "succeed": "succeed"}
_VOWELS = "aeiou"


def _cons(w: str, i: int) -> bool:
    c = w[i]
    if c in _VOWELS:
        return False
    if c == "y":
        return i == 0 or not _cons(w, i - 1)
    return True


def _measure(stem: str) -> int:
    """Number of VC sequences in the [C](VC)^m[V] form of the stem."""
    m, prev_vowel = 0, False
    for i in range(len(stem)):
        if _cons(stem, i):
            if prev_vowel:
                m += 1
            prev_vowel = False
        else:
            prev_vowel = True
    return m


def _has_vowel(stem: str) -> bool:
    return any(not _cons(stem, i) for i in range(len(stem)))


def _double_cons(w: str) -> bool:
    return len(w) >= 2 and w[-1] == w[-2] and _cons(w, len(w) - 1)


def _cvc(w: str) -> bool:
    if len(w) < 3:
        return len(w) == 2 and not _cons(w, 0) and _cons(w, 1)      # nltk extension: a two-letter VC stem counts
    return _cons(w, len(w) - 3) and not _cons(w, len(w) - 2) and _cons(w, len(w) - 1) and w[-1] not in "wxy"


def _replace(w: str, rules, cond) -> str:
    for suf, rep in rules:
        if w.endswith(suf):
            stem = w[:len(w) - len(suf)]
            return stem + rep if cond(stem) else w
    return w


def porter_stem(word: str) -> str:
    w = word.lower()
    if w in _IRREGULAR:
        return _IRREGULAR[w]
    if len(w) <= 2:
        return w
    # step 1a
    if w.endswith("sses"):
        w = w[:-2]
    elif w.endswith("ies"):
        w = w[:-1] if len(w) == 4 else w[:-3] + "i"                 # nltk extension: ties -> tie
    elif w.endswith("ss"):
        pass
    elif w.endswith("s"):
        w = w[:-1]
    # step 1b
    flag = False
    if w.endswith("eed"):
        if _measure(w[:-3]) > 0:
            w = w[:-1]
    elif w.endswith("ied"):                                         # nltk extension: tied -> tie, cried -> cri
        w = w[:-1] if len(w) == 4 else w[:-3] + "i"
    elif w.endswith("ed") and _has_vowel(w[:-2]):
        w, flag = w[:-2], True
    elif w.endswith("ing") and _has_vowel(w[:-3]):
        w, flag = w[:-3], True
    if flag:
        if w.endswith(("at", "bl", "iz")):
            w += "e"
        elif _double_cons(w) and w[-1] not in "lsz":
            w = w[:-1]
        elif _measure(w) == 1 and _cvc(w):
            w += "e"
    # step 1c (nltk extension: y -> i only after a consonant and when the stem is longer than one letter)
    if w.endswith("y") and len(w) > 2 and _cons(w, len(w) - 2):
        w = w[:-1] + "i"
    # step 2 (nltk NLTK_EXTENSIONS form, the stemmer rouge_score uses): "alli" -> "al" is tried FIRST and its result goes through
    # step 2 again ("rationally" -> "rational" -> "ration" ... per nltk/stem/porter.py _step2); the "logi" -> "log" rule measures the
    # stem WITH its "l" (word[:-3]), so that "geology" / "theology" behave like "archaeology"
    def step2(v: str) -> str:
        if v.endswith("alli") and _measure(v[:-4]) > 0:
            return step2(v[:-4] + "al")
        for suf, rep in (("ational", "ate"), ("tional", "tion"), ("enci", "ence"), ("anci", "ance"), ("izer", "ize"), ("bli", "ble"),
                         ("alli", "al"), ("entli", "ent"), ("eli", "e"), ("ousli", "ous"), ("ization", "ize"), ("ation", "ate"),
                         ("ator", "ate"), ("alism", "al"), ("iveness", "ive"), ("fulness", "ful"), ("ousness", "ous"), ("aliti", "al"),
                         ("iviti", "ive"), ("biliti", "ble"), ("fulli", "ful"), ("logi", "log")):
            if v.endswith(suf):
                stem = v[:len(v) - len(suf)]
                ok = _measure(v[:-3]) > 0 if suf == "logi" else _measure(stem) > 0
                return stem + rep if ok else v
        return v
    w = step2(w)
    # step 3
    w = _replace(w, (("icate", "ic"), ("ative", ""), ("alize", "al"), ("iciti", "ic"), ("ical", "ic"), ("ful", ""), ("ness", "")),
                 lambda s: _measure(s) > 0)
    # step 4
    for suf in ("al", "ance", "ence", "er", "ic", "able", "ible", "ant", "ement", "ment", "ent", "ion", "ou", "ism", "ate", "iti",
                "ous", "ive", "ize"):
        if w.endswith(suf):
            stem = w[:len(w) - len(suf)]
            if _measure(stem) > 1 and (suf != "ion" or (stem and stem[-1] in "st")):
                w = stem
            break
    # step 5
    if w.endswith("e"):
        stem = w[:-1]
        m = _measure(stem)
        if m > 1 or (m == 1 and not _cvc(stem)):
            w = stem
    if _measure(w) > 1 and _double_cons(w) and w[-1] == "l":
        w = w[:-1]
    return w


def tokenize(text: str, use_stemmer: bool = True) -> List[str]:
    toks = re.sub(r"[^a-z0-9]+", " ", text.lower()).split()
    if use_stemmer:
        toks = [porter_stem(t) if len(t) > 3 else t for t in toks]
    return [t for t in toks if re.fullmatch(r"[a-z0-9]+", t)]


def _f(p: float, r: float) -> float:
    return 2 * p * r / (p + r) if p + r > 0 else 0.0


def rouge_n(ref: List[str], hyp: List[str], n: int) -> float:
    a = Counter(tuple(ref[i:i + n]) for i in range(len(ref) - n + 1))
    b = Counter(tuple(hyp[i:i + n]) for i in range(len(hyp) - n + 1))
    overlap = sum(min(c, b[g]) for g, c in a.items())
    return _f(overlap / max(sum(b.values()), 1), overlap / max(sum(a.values()), 1))


def rouge_l(ref: List[str], hyp: List[str]) -> float:
    if not ref or not hyp:
        return 0.0
    prev = [0] * (len(hyp) + 1)
    for x in ref:
        cur = [0]
        for j, y in enumerate(hyp):
            cur.append(prev[j] + 1 if x == y else max(prev[j + 1], cur[j]))
        prev = cur
    lcs = prev[-1]
    return _f(lcs / len(hyp), lcs / len(ref))


def mean_rouge_f(reference: str, hypothesis: str, use_stemmer: bool = True) -> float:
    """(rouge1 + rouge2 + rougeL) F-measures / 3, the quantity SG-RLVR.py:158-162 returns."""
    try:
        from rouge_score import rouge_scorer
        s = rouge_scorer.RougeScorer(["rouge1", "rouge2", "rougeL"], use_stemmer=use_stemmer).score(reference, hypothesis)
        return (s["rouge1"].fmeasure + s["rouge2"].fmeasure + s["rougeL"].fmeasure) / 3
    except ImportError:
        ref, hyp = tokenize(reference, use_stemmer), tokenize(hypothesis, use_stemmer)
        return (rouge_n(ref, hyp, 1) + rouge_n(ref, hyp, 2) + rouge_l(ref, hyp)) / 3
