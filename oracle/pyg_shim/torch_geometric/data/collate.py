"""`torch_geometric.data.collate.collate` restated."""
import torch


def collate(cls, data_list, increment=True, add_batch=True, follow_batch=None,
            exclude_keys=None):
    exclude = set(exclude_keys or [])
    out = cls()
    keys = [k for k in data_list[0].keys() if k not in exclude]
    slice_dict, inc_dict = {}, {}
    for key in keys:
        values = [d._store[key] for d in data_list]
        if key == "num_nodes":
            out._store["num_nodes"] = sum(values)
            continue
        if torch.is_tensor(values[0]):
            cat_dim = data_list[0].__cat_dim__(key, values[0])
            sizes = torch.tensor([0] + [v.size(cat_dim) if v.dim() > 0 else 1
                                        for v in values])
            slice_dict[key] = sizes.cumsum(0)
            if increment:
                incs, run = [], 0
                for d, v in zip(data_list, values):
                    incs.append(run)
                    step = d.__inc__(key, v)
                    run = run + (int(step) if not isinstance(step, int) else step)
                inc_dict[key] = torch.tensor(incs)
                if any(i != 0 for i in incs):
                    values = [v + i for v, i in zip(values, incs)]
            if values[0].dim() == 0:
                out._store[key] = torch.stack(values)
            else:
                out._store[key] = torch.cat(values, dim=cat_dim)
        else:
            out._store[key] = values
    if add_batch:
        n = [int(d.num_nodes) for d in data_list]
        dev = None
        for d in data_list:
            for v in d._store.values():
                if torch.is_tensor(v):
                    dev = v.device
                    break
            break
        out._store["batch"] = torch.repeat_interleave(
            torch.arange(len(n), device=dev), torch.tensor(n, device=dev))
        out._store["ptr"] = torch.tensor([0] + n, device=dev).cumsum(0)
    return out, slice_dict, inc_dict
