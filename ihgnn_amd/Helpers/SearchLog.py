"""Search-log rows of the training / validation / test CSVs (reference ``Helpers/SearchLog.py:3-75,184-207``).

Row format (8 comma-separated columns, inner lists space-separated):
``user,query,search_time,items,pages,positions,interactions,times``.
"""
from typing import Dict, List, NamedTuple, Tuple

COLUMNS = ('user', 'query', 'search_time', 'items', 'pages', 'positions', 'interactions', 'times')


def _ints(field: str) -> List[int]:
    return [int(tok) for tok in field.split()]


class SearchLog(NamedTuple):
    user: int
    query: int
    search_time: str
    items: List[int] = None
    pages: List[int] = None
    positions: List[int] = None
    interactions: List[int] = None
    times: List[str] = None

    @classmethod
    def parse(cls, s: str) -> 'SearchLog':
        fields = s.strip().split(',')
        if len(fields) != len(COLUMNS):
            raise ValueError(f'search-log row needs {len(COLUMNS)} columns, got {len(fields)}: {s!r}')
        user, query, stamp, items, pages, positions, flags, times = fields
        return cls(int(user), int(query), stamp, _ints(items), _ints(pages), _ints(positions), _ints(flags), times.split())

    def tostr(self) -> str:
        join = lambda seq: ' '.join(str(x) for x in seq)
        return ','.join([str(self.user), str(self.query), self.search_time, join(self.items), join(self.pages),
                         join(self.positions), join(self.interactions), join(self.times)])

    __str__ = tostr

    @staticmethod
    def column_names() -> str:
        return ','.join(COLUMNS)

    def get_interacted_items(self, flag_policy: str = 'min') -> Tuple[List[int], List[int], bool]:
        """Distinct positively-interacted items (first-seen order), one relevance each, and "all are 1"."""
        pick = min if flag_policy == 'min' else max
        seen: Dict[int, int] = {}
        for item, flag in zip(self.items, self.interactions):
            if flag > 0:
                seen[item] = flag if item not in seen else pick(seen[item], flag)
        items, flags = list(seen.keys()), list(seen.values())
        return items, flags, all(f <= 1 for f in flags)


class PosInteraction(NamedTuple):
    user: int
    query: int
    search_time: str
    item: int
    page: int
    position: int
    interaction: int
    time: str

    def uqif(self) -> Tuple[int, int, int, int]:
        return self.user, self.query, self.item, self.interaction

    def uqift(self) -> Tuple[int, int, int, int, str]:
        return self.user, self.query, self.item, self.interaction, self.time

    @staticmethod
    def from_search_log(log: SearchLog, treat_all_1: bool) -> List['PosInteraction']:
        out = []
        for item, page, pos, flag, when in zip(log.items, log.pages, log.positions, log.interactions, log.times):
            if flag > 0:
                out.append(PosInteraction(log.user, log.query, log.search_time, item, page, pos,
                                          1 if treat_all_1 else flag, when))
        return out
