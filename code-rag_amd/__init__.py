"""coderag_amd -- MI355X-native embed-and-search hot path of iAmLakshya/code-rag (`lattice`).

Import name: ``coderag_amd`` (the directory is ``code-rag_amd/``; ``coderag_amd.py`` at the repo
root maps one onto the other).  Only the path SURVEY.md section 8 names lives here: the HBM-resident
cosine index behind the reference's ``QdrantManager`` surface, the UniXcoder encoder behind its
``EmbeddingProvider`` surface, the semantic-search and ranking surfaces that call them, and the
HIP kernels + C ABI underneath (``csrc/``, ``include/coderag_hip.h``).
"""

__version__ = "0.1.0"
