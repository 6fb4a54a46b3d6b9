"""A list of :class:`SearchLog` rows with CSV read/write (reference ``Helpers/SearchLogCollection.py:6-32``)."""
from typing import Iterator, List

from .SearchLog import SearchLog


class SearchLogCollection:
    def __init__(self, logs=()):
        self.logs: List[SearchLog] = list(logs)

    def __getitem__(self, index: int) -> SearchLog:
        return self.logs[index]

    def __len__(self) -> int:
        return len(self.logs)

    def __iter__(self) -> Iterator[SearchLog]:
        return iter(self.logs)

    def append(self, log: SearchLog) -> None:
        self.logs.append(log)

    def write(self, filename: str, encoding: str = 'utf-8') -> None:
        with open(filename, 'w', encoding=encoding) as f:
            f.write(SearchLog.column_names() + '\n')
            f.writelines(log.tostr() + '\n' for log in self.logs)

    @classmethod
    def read(cls, filename: str, encoding: str = 'utf-8') -> 'SearchLogCollection':
        with open(filename, 'r', encoding=encoding) as f:
            next(f, None)                                   # header
            return cls(SearchLog.parse(line) for line in f if line.strip())
