"""numpy restatement of the reference's edge-embedding operators
(embiggen/embedding_transformers/edge_transformer.py:12-343).  TEST INFRASTRUCTURE ONLY.

Unlike the training path this part of the reference IS executable here: the functions are plain
numpy, so tests/golden/edge_embedding_cases.npz holds outputs of the reference's own functions
(made by tests/golden/make_reference_fixtures.py) and this restatement is pinned against them.
"""
import numpy as np


def l2_norm(e):
    return np.sqrt(np.power(e, 2.0).sum(axis=1, keepdims=True))  # :176-191


def cosine(s, d):
    norm = l2_norm(s) * l2_norm(d)
    norm[norm < 1e-6] = 1e-6  # :266
    return np.sum(s * d, axis=1, keepdims=True) / norm


METHODS = {
    "Hadamard": lambda s, d: s * d,                                   # :12-32
    "Sum": lambda s, d: s + d,                                        # :35-55
    "Average": lambda s, d: (s + d) / 2.0,                            # :58-81
    "L1": lambda s, d: s - d,                                         # :84-104
    "AbsoluteL1": lambda s, d: np.abs(s - d),                         # :125-147
    "SquaredL2": lambda s, d: np.power(s - d, 2.0),                   # :150-173
    "L2": lambda s, d: np.sqrt(np.power(s - d, 2.0)),                 # :194-216
    "Concatenate": lambda s, d: np.hstack((s, d)),                    # :270-290
    "Min": lambda s, d: np.minimum(s, d),                             # :293-316
    "Max": lambda s, d: np.maximum(s, d),                             # :319-342
    "L2Distance": lambda s, d: np.sqrt(np.sum(np.power(s - d, 2.0), axis=1)).reshape((-1, 1)),
    "CosineSimilarity": cosine,                                       # :242-267
}


def edge_embedding(table, sources, destinations, method, destination_table=None):
    dst_table = table if destination_table is None else destination_table
    return METHODS[method](table[sources], dst_table[destinations])
