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

With stock BM25 similarities on whole (unsliced) arrays the combination itself runs on the GPU
(``_DeviceCombiner``: Part 4 of the C ABI, csrc/sa_vec.hip): every per-field, per-term score is computed
straight into a device vector and only the final result crosses PCIe.  The arithmetic follows numpy's
operation for operation, so both routes return identical arrays (tests/test_solr.py runs each scenario
through both).  Anything else -- custom similarity callables, slices -- takes the host route.
"""
from __future__ import annotations

import ctypes
import re
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import pandas as pd

from .postings import SearchArray
from .similarity import Similarity, default_bm25

_INT_ERR = "Invalid 'mm' spec. Expecting an integer."
_F32P = ctypes.POINTER(ctypes.c_float)


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


class _DeviceCombiner:
    """The edismax combination on device vectors; mirrors _term_centric / _field_centric / _phrase_phase."""

    def __init__(self, fields: List[_Field], n_docs: int):
        from .device_index import DeviceVec
        self.fields = fields
        self.n = n_docs
        self.devs = {f.name: f.array._core.device() for f in fields}
        self.api = next(iter(self.devs.values())).api
        self._Vec = DeviceVec
        self._vecs = []

    @staticmethod
    def usable(fields: List[_Field], n_docs: int) -> bool:
        if n_docs == 0 or not fields:
            return False
        for f in fields:
            sim = f.similarity
            if getattr(sim, "kind", None) != "bm25" or sim.k1 == 0 or sim.b == 1:
                return False
            # a negative boost makes the main-query scores mixed-sign; the reference then slices with
            # `scores > 0` but adds at `np.where(scores)` and raises a broadcast ValueError
            # (solr.py:320-353): only the host route reproduces that
            if f.boost is not None and f.boost < 0:
                return False
            if f.array._rows is not None or len(f.array) != n_docs or len(f.array._core.doc_lens) == 0:
                return False
        return True

    def vec(self, f64: bool):
        v = self._Vec(self.api, self.n, f64)
        self._vecs.append(v)
        return v

    def close(self):
        for v in self._vecs:
            v.close()
        self._vecs = []

    # one term or phrase of one field -> float32 vector (times boost), idf from the given docfreqs
    def score_into(self, out, f: _Field, tokens: List[str], boost, dfs=None):
        from .device_index import NO_TERM, compute_idf, p_u32
        arr, dev = f.array, self.devs[f.name]
        ids = [arr._term_id(t) for t in tokens]
        tarr = np.asarray([i if i >= 0 else NO_TERM for i in ids], dtype=np.uint32)
        if dfs is None:
            dfs = np.asarray([arr.docfreq(t) for t in tokens])
        idf = np.float32(compute_idf(arr.corpus_size, np.asarray(dfs)))
        k1, b = np.float32(f.similarity.k1), np.float32(f.similarity.b)
        if len(tokens) == 1:
            dev.into_vec(out, boost, "sa_index_bm25_dense", p_u32(tarr), np.asarray([idf], np.float32).ctypes.data_as(_F32P), 1, k1, b)
        else:
            dev.into_vec(out, boost, "sa_index_bm25_phrase_dense_posn", p_u32(tarr), len(tarr), 0, -1, -1, idf, k1, b)

    def term_centric(self, n_terms: int, need: int, tie: float):
        call = self.api.call
        total, cnt = self.vec(True), self.vec(False)
        acc, best, s = self.vec(True), self.vec(True), self.vec(False)
        for posn in range(n_terms):
            acc.zero()
            best.zero()
            for f in self.fields:
                self.score_into(s, f, [f.terms[posn]], f.boost)
                call("sa_vec_dismax_acc", acc._h, best._h, s._h)
            call("sa_vec_clause", acc._h, best._h, float(tie), total._h, cnt._h)
        call("sa_vec_mask_min_count", total._h, cnt._h, int(need))
        return total

    def field_centric(self, mm: str, tie: float):
        call = self.api.call
        fsum, fmax, out = self.vec(False), self.vec(False), self.vec(False)
        summed, cnt, s = self.vec(False), self.vec(False), self.vec(False)
        for k, f in enumerate(self.fields):
            summed.zero()
            cnt.zero()
            for t in f.terms:
                self.score_into(s, f, [t], None)
                call("sa_vec_sum_count32", summed._h, cnt._h, s._h)
            need = min(parse_min_should_match(len(f.terms), spec=mm), len(f.terms))
            call("sa_vec_field_row", summed._h, cnt._h, int(need), np.float32(_weight(f.boost)), 0 if f.boost is None else 1,
                 1 if k == 0 else 0, fsum._h, fmax._h)
        call("sa_vec_field_finish", fsum._h, fmax._h, np.float32(tie), out._h)
        return out

    def phrase_phase(self, scores, mask0, per_field, boosts, n):
        """scores[matched] += sum of the boosted phrase scores computed with the matched docs' docfreqs"""
        from .device_index import NO_TERM
        call = self.api.call
        by_name = {f.name: f for f in self.fields}
        extra, piece, tf = self.vec(False), self.vec(False), self.vec(False)
        n_pieces = 0
        for field, boost in boosts.items():
            terms = per_field[field]
            if len(terms) < (2 if n is None else n):
                continue
            f = by_name[field]                       # KeyError for a phrase field outside qf, as in the reference
            dev = self.devs[field]
            grams = [terms] if n is None else _ngrams(terms, n)
            if n == 2:
                grams = grams + [grams[-1]]          # reference solr.py:217: the field's last bigram counts twice
            for gram in grams:
                dfs = []
                for tok in gram:                     # docfreq among the matched docs (slices are subset-local)
                    tid = f.array._term_id(tok)
                    c = ctypes.c_uint64(0)
                    if tid >= 0:
                        dev.into_vec(tf, None, "sa_index_termfreqs_dense_posn", tid, -1, -1)
                        call("sa_vec_count_where", tf._h, mask0._h, ctypes.byref(c))
                    dfs.append(c.value)
                self.score_into(piece, f, list(gram), boost, dfs=dfs)
                call("sa_vec_add32", extra._h, piece._h, 1 if n_pieces == 0 else 0)
                n_pieces += 1
        if n_pieces:
            call("sa_vec_add_where", scores._h, extra._h, scores._h)


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
            similarity: Union[Similarity, Dict[str, Similarity]] = default_bm25,
            use_device: Optional[bool] = None) -> Tuple[np.ndarray, str]:
    """Solr's extended-dismax over a dataframe whose ``qf`` columns are ``SearchArray`` s.

    q: query string; qf: fields with optional ``^boost``; mm: minimum-should-match spec (default
    ``"1"``; ``q_op="AND"`` forces ``"100%"``); pf / pf2 / pf3: fields boosted when the whole query /
    its bigrams / its trigrams match as a phrase; tie: dismax tie breaker; similarity: one callable or
    one per field.  ps / ps2 / ps3 are accepted for signature parity and, as in the reference, unused.
    use_device: None picks the GPU combination whenever it applies (stock BM25, unsliced arrays); False
    forces the host route.  Returns ``(scores[len(frame)], explain string)`` -- float64 term-centric,
    float32 field-centric, as the reference."""
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
    phases = ((phrase_boosts, None), (bigram_boosts, 2), (trigram_boosts, 3))
    if use_device is None:
        use_device = _DeviceCombiner.usable(fields, len(frame))
    if use_device:
        # everything stays in HBM until the final vector; the explain string is the host route's
        comb = _DeviceCombiner(fields, len(frame))
        try:
            if term_centric:
                need = parse_min_should_match(n_terms, spec=mm)
                scores_v = comb.term_centric(n_terms, need, tie)
            else:
                scores_v = comb.field_centric(mm, tie)
            explain = _explain_main(fields, n_terms, mm, term_centric)
            mask0 = comb.vec(scores_v.f64)
            mask0.copy_from(scores_v)
            for boosts, n in phases:
                comb.phrase_phase(scores_v, mask0, per_field, boosts, n)
                explain += _explain_phase(per_field, boosts, n)
            return scores_v.fetch(), explain
        finally:
            comb.close()

    if term_centric:
        scores, explain = _term_centric(fields, len(frame), n_terms, mm, tie)
    else:
        scores, explain = _field_centric(fields, len(frame), mm, tie)

    # phrase boosts only look at (and only add to) the docs the main query matched
    hit = scores > 0
    matched = {f.name: f.array[hit] for f in fields}
    for boosts, n in phases:
        extra, text = _phrase_phase(matched, per_field, boosts, similarity, n)
        explain += text
        if extra is not None:
            scores[np.where(scores)[0]] += extra
    return scores, explain


def _explain_main(fields: List[_Field], n_terms: int, mm: str, term_centric: bool) -> str:
    if term_centric:
        clauses = ["(" + " | ".join(f"{f.name}:{f.terms[p]}^{_boost_text(f.boost)}" for f in fields) + ")" for p in range(n_terms)]
        return "(" + " ".join(clauses) + f")~{parse_min_should_match(n_terms, spec=mm)}"
    texts = []
    for f in fields:
        need = min(parse_min_should_match(len(f.terms), spec=mm), len(f.terms))
        clause = " ".join(f"{f.name}:{t}" for t in f.terms)
        texts.append(f"(({clause})~{need})^{_boost_text(f.boost)}")
    return " | ".join(texts)


def _explain_phase(per_field, boosts, n) -> str:
    text = ""
    for field, boost in boosts.items():
        terms = per_field[field]
        if len(terms) < (2 if n is None else n):
            continue
        for gram in ([terms] if n is None else _ngrams(terms, n)):
            text += f" ({field}:\"{' '.join(gram)}\")^{_boost_text(boost)}"
    return text
