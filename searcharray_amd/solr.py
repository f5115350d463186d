"""Solr-style query helpers on top of ``SearchArray`` columns: ``edismax`` and the ``mm`` grammar.

Same public names, arguments and results as the reference's ``searcharray/solr.py`` (v0.0.73) --
``parse_min_should_match`` (solr.py:10-60), ``parse_field_boosts`` (:63-75), ``edismax`` (:239-355) --
so code written against the reference runs unchanged; the scoring calls underneath are
``SearchArray.score`` on the GPU.  The combination rules, including two behaviours that look
accidental but are observable, are the reference's:

* term-centric when every query field tokenises the query into the same number of terms
  (solr.py:87-109, 112-144), field-centric otherwise (:147-176);
* phrase boosts (pf / pf2 / pf3) are computed on the docs the main query matched and added there
  (:320-353); in the pf2 phase the LAST bigram of every field is added twice (:210-218).
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import pandas as pd

from .postings import SearchArray
from .similarity import Similarity, default_bm25

_INT_ERR = "Invalid 'mm' spec. Expecting an integer."


def _mm_int(text: str) -> int:
    try:
        return int(text)
    except ValueError:
        raise ValueError(_INT_ERR)


def _mm_simple(num_clauses: int, result: int, spec: str) -> int:
    """``N``, ``-N``, ``P%`` or ``-P%`` applied to ``result`` clauses (reference solr.py:47-60)."""
    if "%" in spec:
        share = (result * _mm_int(spec[:-1])) * (1 / 100)       # '%' is assumed to be the last character
        result = result + int(share) if share < 0 else int(share)
    else:
        n = _mm_int(spec)
        result = result + n if n < 0 else n
    return min(num_clauses, max(result, 0))


def parse_min_should_match(num_clauses: int, spec: str) -> int:
    """Solr's ``mm`` ("minimum should match") grammar: how many of ``num_clauses`` optional clauses a
    document must match.  ``"3"``, ``"-2"``, ``"75%"``, ``"-25%"`` and conditional lists such as
    ``"2<-25% 9<-3"`` (the rule after the largest bound below ``num_clauses`` applies)."""
    spec = spec.strip()
    if "<" not in spec:
        return _mm_simple(num_clauses, num_clauses, spec)
    required = num_clauses
    for rule in re.sub(r"\s*<\s*", "<", spec).split():
        bound, sep, then = rule.partition("<")
        if not sep:
            raise ValueError("Invalid 'mm' spec: '" + rule + "'. Expecting values before and after '<'")
        if num_clauses <= _mm_int(bound):
            return required
        required = parse_min_should_match(num_clauses, then)
    return required


def parse_field_boosts(field_lists: Optional[Sequence[str]]) -> Dict[str, Optional[float]]:
    """``["title^10", "body"]`` -> ``{"title": 10.0, "body": None}`` (qf, pf, pf2, pf3)."""
    boosts: Dict[str, Optional[float]] = {}
    for spec in field_lists or []:
        pieces = spec.split("^")
        boosts[pieces[0]] = float(pieces[1]) if len(pieces) > 1 else None
    return boosts


def get_field(frame: pd.DataFrame, field: str) -> SearchArray:
    if field not in frame.columns:
        raise ValueError(f"Field {field} not in dataframe")
    column = frame[field].array
    if not isinstance(column, SearchArray):
        raise ValueError(f"Field {field} is not a searcharray field")
    return column


def _weight(boost: Optional[float]) -> float:
    return 1 if boost is None else boost


def _boost_text(boost: Optional[float]) -> str:
    return "1" if boost is None else f"{boost}"


@dataclass
class _Field:
    name: str
    boost: Optional[float]
    array: SearchArray
    terms: List[str]
    similarity: Similarity


def parse_query_terms(frame: pd.DataFrame, query: str, query_fields: Sequence[str]):
    """Tokenise ``query`` with every field's own tokenizer.  Term-centric scoring needs the same
    number of terms from every field."""
    per_field = {f: list(get_field(frame, f).tokenizer(query)) for f in query_fields}
    counts = [len(t) for t in per_field.values()]
    n_terms = next((c for c in counts if c), 0)
    # the reference compares every later field with the first non-zero count (solr.py:103-107)
    term_centric = True
    seen = 0
    for c in counts:
        if seen == 0:
            seen = c
        elif c != seen:
            term_centric = False
    return n_terms, per_field, term_centric


def _dismax(per_field_scores: np.ndarray, tie: float) -> np.ndarray:
    """disjunction-max over axis 0 with a tie breaker: max + tie * (sum - max)"""
    best = per_field_scores.max(axis=0)
    return best + (per_field_scores.sum(axis=0) - best) * tie


def _term_centric(fields: List[_Field], n_docs: int, n_terms: int, mm: str, tie: float) -> Tuple[np.ndarray, str]:
    clause_scores = np.zeros((n_terms, n_docs), dtype=np.float64)
    clauses = []
    for posn in range(n_terms):
        # float64 accumulators like the reference's np.zeros(len(frame)) (solr.py:124-125)
        stack = np.zeros((len(fields) + 1, n_docs), dtype=np.float64)
        parts = []
        for i, f in enumerate(fields):
            term = f.terms[posn]
            stack[i] = f.array.score(term, similarity=f.similarity) * _weight(f.boost)
            parts.append(f"{f.name}:{term}^{_boost_text(f.boost)}")
        total = np.zeros(n_docs)
        for i in range(len(fields)):
            total += stack[i]                                   # same left-to-right order as the reference
        best = stack.max(axis=0)
        clause_scores[posn] = best + (total - best) * tie
        clauses.append("(" + " | ".join(parts) + ")")
    need = parse_min_should_match(n_terms, spec=mm)
    enough = (clause_scores > 0).sum(axis=0) >= need
    scores = np.sum(list(clause_scores), axis=0) if n_terms else np.zeros(n_docs)
    scores[~enough] = 0
    return scores, "(" + " ".join(clauses) + f")~{need}"


def _field_centric(fields: List[_Field], n_docs: int, mm: str, tie: float) -> Tuple[np.ndarray, str]:
    rows = []
    texts = []
    for f in fields:
        per_term = np.array([f.array.score(t, similarity=f.similarity) for t in f.terms])
        need = min(parse_min_should_match(len(f.terms), spec=mm), len(f.terms))
        enough = np.sum(per_term > 0, axis=0) >= need
        summed = np.sum(per_term, axis=0)
        summed[~enough] = 0
        rows.append(summed * _weight(f.boost))
        clause = " ".join(f"{f.name}:{t}" for t in f.terms)
        texts.append(f"(({clause})~{need})^{_boost_text(f.boost)}")
    return _dismax(np.asarray(rows), tie), " | ".join(texts)


def _ngrams(terms: List[str], n: int):
    return [terms[i:i + n] for i in range(len(terms) - n + 1)]


def _phrase_phase(matched: Dict[str, SearchArray], per_field: Dict[str, List[str]], boosts: Dict[str, Optional[float]],
                  similarity: Dict[str, Similarity], n: Optional[int]):
    """Sum of the boosted phrase scores over the matched docs.  n=None: the whole query as one phrase
    (pf, needs >= 2 terms); n=2 / n=3: every bigram / trigram (pf2 / pf3)."""
    pieces: List[np.ndarray] = []
    text = ""
    for field, boost in boosts.items():
        terms = per_field[field]
        if len(terms) < (2 if n is None else n):
            continue
        grams = [terms] if n is None else _ngrams(terms, n)
        last = None
        for gram in grams:
            last = matched[field].score(list(gram), similarity=similarity[field]) * _weight(boost)
            text += f" ({field}:\"{' '.join(gram)}\")^{_boost_text(boost)}"
            pieces.append(last)
        if n == 2 and last is not None:
            pieces.append(last)            # reference solr.py:217: the field's last bigram counts twice
    return (np.sum(pieces, axis=0) if pieces else None), text


def edismax(frame: pd.DataFrame,
            q: str,
            qf: List[str],
            mm: Optional[Union[str, int]] = None,
            pf: Optional[List[str]] = None,
            pf2: Optional[List[str]] = None,
            pf3: Optional[List[str]] = None,
            ps2: int = 0,
            ps3: int = 0,
            ps: int = 0,
            tie: float = 0.0,
            q_op: str = "OR",
            similarity: Union[Similarity, Dict[str, Similarity]] = default_bm25) -> Tuple[np.ndarray, str]:
    """Solr's extended-dismax over a dataframe whose ``qf`` columns are ``SearchArray`` s.

    q: query string; qf: fields with optional ``^boost``; mm: minimum-should-match spec (default
    ``"1"``; ``q_op="AND"`` forces ``"100%"``); pf / pf2 / pf3: fields boosted when the whole query /
    its bigrams / its trigrams match as a phrase; tie: dismax tie breaker; similarity: one callable or
    one per field.  ps / ps2 / ps3 are accepted for signature parity and, as in the reference, unused.
    Returns ``(scores float64[len(frame)], explain string)``."""
    def as_list(x):
        return x if isinstance(x, list) else [x]

    query_boosts = parse_field_boosts(as_list(qf))
    phrase_boosts = parse_field_boosts(as_list(pf)) if pf else {}
    bigram_boosts = parse_field_boosts(pf2) if pf2 else {}
    trigram_boosts = parse_field_boosts(pf3) if pf3 else {}
    mm = "1" if mm is None else (f"{mm}" if isinstance(mm, int) else mm)
    if q_op == "AND":
        mm = "100%"
    if not isinstance(similarity, dict):
        similarity = {field: similarity for field in query_boosts}
    for field in query_boosts:
        similarity.setdefault(field, default_bm25)

    n_terms, per_field, term_centric = parse_query_terms(frame, q, list(query_boosts))
    fields = [_Field(name, boost, get_field(frame, name), per_field[name], similarity[name])
              for name, boost in query_boosts.items()]
    if term_centric:
        scores, explain = _term_centric(fields, len(frame), n_terms, mm, tie)
    else:
        scores, explain = _field_centric(fields, len(frame), mm, tie)

    # phrase boosts only look at (and only add to) the docs the main query matched
    hit = scores > 0
    matched = {f.name: f.array[hit] for f in fields}
    for boosts, n in ((phrase_boosts, None), (bigram_boosts, 2), (trigram_boosts, 3)):
        extra, text = _phrase_phase(matched, per_field, boosts, similarity, n)
        explain += text
        if extra is not None:
            scores[np.where(scores)[0]] += extra
    return scores, explain
