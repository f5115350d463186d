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
            tid = len(self._terms)
            self._ids[term] = tid
            self._terms.append(term)
        return tid

    def get_term_id(self, term: str) -> int:
        try:
            return self._ids[term]
        except KeyError:
            raise TermMissingError(f"Term {term} not present in dictionary. Reindex to add.")

    def get_term(self, term_id: int) -> str:
        if 0 <= term_id < len(self._terms):
            return self._terms[term_id]
        raise TermMissingError(f"Term at {term_id} not present in dictionary. Reindex to add.")

    def copy(self) -> "TermDict":
        other = TermDict()
        other._ids = dict(self._ids)
        other._terms = list(self._terms)
        return other

    def compatible(self, other: "TermDict") -> bool:
        n = min(len(self._terms), len(other._terms))
        return self._terms[:n] == other._terms[:n]

    def __len__(self) -> int:
        return len(self._terms)

    def __contains__(self, term: str) -> bool:
        return term in self._ids

    def __repr__(self) -> str:
        return repr(self._ids)

    @property
    def nbytes(self) -> int:
        return sys.getsizeof(self._ids) + sys.getsizeof(self._terms)
