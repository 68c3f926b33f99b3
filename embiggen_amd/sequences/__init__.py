"""Batch generators of the gn2v engine (drop-in names of ``embiggen.sequences``)."""
from .node2vec_sequence import Node2VecSequence

__all__ = ["Node2VecSequence"]
