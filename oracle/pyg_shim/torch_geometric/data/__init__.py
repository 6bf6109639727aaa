"""`Data` / `Batch` / `Dataset` restated from SURVEY App. A-5.

`Data` is an attribute bag; `collate` concatenates tensors along dim 0 (keys
containing 'index' along the last dim), increments keys containing 'index' by
the running node count and keys containing 'batch' by max+1, sums
`num_nodes`, and (add_batch) writes the graph-id vector `batch` and `ptr`."""
import torch
from .collate import collate  # noqa: F401  (re-export, data.py:8 imports the submodule)


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None,
                 **kwargs):
        self.__dict__["_store"] = {}
        for k, v in dict(x=x, edge_index=edge_index, edge_attr=edge_attr, y=y,
                         pos=pos).items():
            if v is not None:
                self._store[k] = v
        for k, v in kwargs.items():
            self._store[k] = v

    def __getattr__(self, key):
        store = self.__dict__.get("_store", {})
        if key in store:
            return store[key]
        raise AttributeError(key)

    def __setattr__(self, key, value):
        self._store[key] = value

    def __contains__(self, key):
        return key in self._store

    def keys(self):
        return list(self._store.keys())

    def __inc__(self, key, value):
        if "batch" in key:
            return int(value.max()) + 1
        if "index" in key or "face" in key:
            return self.num_nodes
        return 0

    def __cat_dim__(self, key, value):
        return -1 if ("index" in key or "face" in key) else 0

    def to(self, device, *args, **kwargs):
        for k, v in list(self._store.items()):
            if torch.is_tensor(v):
                self._store[k] = v.to(device, *args, **kwargs)
        return self


class Batch(Data):
    @classmethod
    def from_data_list(cls, data_list, follow_batch=None, exclude_keys=None):
        batch, _, _ = collate(cls, data_list=data_list, increment=True,
                              add_batch=True, exclude_keys=exclude_keys)
        return batch


class Dataset(torch.utils.data.Dataset):
    def __init__(self, root=None, transform=None, pre_transform=None, pre_filter=None):
        super().__init__()

    def __len__(self):
        return self.len()

    def __getitem__(self, idx):
        return self.get(idx)
