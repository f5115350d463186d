"""str <-> int term dictionary (host side).  Same contract as the reference's
searcharray/term_dict.py:4-59: ids are dense and assigned in first-seen order, unknown terms
raise TermMissingError (a KeyError), `compatible` compares the common prefix of two dictionaries."""
from __future__ import annotations

import sys
from typing import Dict, List


class TermMissingError(KeyError):
    pass


class TermDict:
    def __init__(self):
        self._ids: Dict[str, int] = {}
        self._terms: List[str] = []

    def add_term(self, term: str) -> int:
        tid = self._ids.get(term)
        if tid is None:
            self._sync()
            tid = len(self._terms)
            self._ids[term] = tid
            self._terms.append(term)
        return tid

    def add_terms(self, terms) -> List[int]:
        """ids of a doc's tokens, adding unseen ones in first-seen order.  One C-level dict call per
        token (the hot loop of index build); the id -> term list is rebuilt lazily from the dict's
        insertion order."""
        ids = self._ids
        setdefault = ids.setdefault
        return [setdefault(t, len(ids)) for t in terms]

    def _sync(self):
        if len(self._terms) != len(self._ids):
            self._terms = list(self._ids)           # dicts keep insertion order == id order

    def get_term_id(self, term: str) -> int:
        try:
            return self._ids[term]
        except KeyError:
            raise TermMissingError(f"Term {term} not present in dictionary. Reindex to add.")

    def get_term(self, term_id: int) -> str:
        self._sync()
        if 0 <= term_id < len(self._terms):
            return self._terms[term_id]
        raise TermMissingError(f"Term at {term_id} not present in dictionary. Reindex to add.")

    def copy(self) -> "TermDict":
        self._sync()
        other = TermDict()
        other._ids = dict(self._ids)
        other._terms = list(self._terms)
        return other

    def compatible(self, other: "TermDict") -> bool:
        self._sync()
        other._sync()
        n = min(len(self._terms), len(other._terms))
        return self._terms[:n] == other._terms[:n]

    def __len__(self) -> int:
        return len(self._ids)

    def __contains__(self, term: str) -> bool:
        return term in self._ids

    def __repr__(self) -> str:
        return repr(self._ids)

    @property
    def nbytes(self) -> int:
        self._sync()
        return sys.getsizeof(self._ids) + sys.getsizeof(self._terms)
