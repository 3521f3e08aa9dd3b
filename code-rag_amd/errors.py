"""Error hierarchy of the hot path -- same class names, constructor arguments and ``str()`` form as the
reference's ``src/lattice/core/errors.py:1-76`` (only the classes the path raises)."""

from __future__ import annotations


class CodeRAGError(Exception):
    """Base error; ``cause`` carries the underlying exception (core/errors.py:1-10)."""

    def __init__(self, message: str, cause: Exception | None = None):
        super().__init__(message)
        self.cause = cause

    def __str__(self) -> str:
        base = self.args[0]
        return f"{base} (caused by: {self.cause})" if self.cause else str(base)


class ConfigurationError(CodeRAGError):
    pass


class VectorStoreError(CodeRAGError):
    pass


class EmbeddingError(CodeRAGError):
    pass


class QueryError(CodeRAGError):
    pass


class IndexingError(CodeRAGError):
    """core/errors.py:45-53: carries the pipeline ``stage`` that failed."""

    def __init__(self, message: str, stage: str | None = None, cause: Exception | None = None):
        super().__init__(message, cause)
        self.stage = stage
