"""ihgnn_amd - MI355X-native hypergraph message passing for IHGNN-style personalised product search.

Importing the package is cheap and GPU-free; the HIP library (``ihgnn_amd/csrc/libihgnn_hip.so``)
is bound on first use by :mod:`ihgnn_amd._lib` and its absence is a hard error, never a fallback.
"""
__version__ = '0.1.0'
