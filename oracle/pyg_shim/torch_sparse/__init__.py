"""Import stub (oracle only). model.py:6 imports these names; they are reached
only through isinstance checks / a dead branch (model.py:31-34, 74-75)."""


class SparseTensor:  # never instantiated on the captured path
    pass


def masked_select_nnz(*args, **kwargs):
    raise NotImplementedError("torch_sparse is not available; dead branch in model.py:34")
