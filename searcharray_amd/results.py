"""``SetOfResults``: collect the top rows of several queries over one dataframe, e.g. to eyeball or
export a judgement list.  Same interface as the reference's ``searcharray.utils.sort.SetOfResults``
(utils/sort.py:9-52): ``ins_top_n(scores, N, query, metadata)`` then ``get_all()``."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np
import pandas as pd

from .postings import SearchArray


class SetOfResults:
    def __init__(self, df: pd.DataFrame, searchable: bool = False):
        self.df = df
        self.indices: List[int] = []
        self.metadata: Dict[str, List[Any]] = {}
        # SearchArray columns are left out of the result frame unless asked for
        self.searchable_cols = [] if searchable else [c for c in df.columns if isinstance(df[c].array, SearchArray)]

    def _extend(self, key: str, values) -> None:
        self.metadata.setdefault(key, []).extend(values)

    def ins_top_n(self, scores, N: int = 10, query: str = "", metadata: Optional[Dict[str, Any]] = None) -> None:
        """Record the N best-scoring rows of ``scores`` under ``query`` (any order among them, like the
        reference's argpartition; ``get_all`` ranks them)."""
        scores = np.asarray(scores)
        top = np.argpartition(scores, -N)[-N:]
        self.indices.extend(int(i) for i in top)
        self._extend("score", scores[top])
        self._extend("query", [query] * len(top))
        for key, values in (metadata or {}).items():
            self._extend(key, values if isinstance(values, list) else [values] * len(top))
            if len(self.metadata[key]) != len(self.indices):
                raise ValueError("Metadata must have same length as scores.")

    def get_all(self) -> pd.DataFrame:
        keep = [c for c in self.df.columns if c not in self.searchable_cols]
        out = self.df.iloc[self.indices][keep].copy()
        for key, values in self.metadata.items():
            out[key] = values
        out = out.sort_values(["query", "score"], ascending=[True, False])
        out["rank"] = out.groupby("query").cumcount() + 1
        return out.reset_index(drop=True)
