"""`RGCNConv` skeleton + `MessagePassing.propagate`.

Only what model.py:41-135 relies on: parameter registration order
(weight, root, bias), glorot/zeros init, `in_channels_l`, and
propagate = gather x[src] -> message() -> scatter-mean onto dst."""
import inspect
import torch
from torch import Tensor
from torch.nn import Parameter
from .inits import glorot, zeros


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim
        self._msg_params = [p for p in inspect.signature(self.message).parameters]

    def propagate(self, edge_index, size=None, **kwargs):
        src, dst = edge_index[0], edge_index[1]
        call = {}
        for name in self._msg_params:
            if name.endswith("_j"):
                call[name] = kwargs[name[:-2]].index_select(0, src)
            elif name.endswith("_i"):
                call[name] = kwargs[name[:-2]].index_select(0, dst)
            else:
                call[name] = kwargs[name]
        msg = self.message(**call)
        n_out = size[1] if size is not None else int(kwargs["x"].size(0))
        out = torch.zeros((n_out,) + tuple(msg.shape[1:]), dtype=msg.dtype,
                          device=msg.device).index_add_(0, dst, msg)
        if self.aggr == "mean":
            cnt = torch.zeros(n_out, dtype=msg.dtype, device=msg.device)
            cnt.index_add_(0, dst, torch.ones(dst.numel(), dtype=msg.dtype,
                                              device=msg.device))
            out = out / cnt.clamp(min=1).view(-1, *([1] * (msg.dim() - 1)))
        elif self.aggr not in ("add", "sum"):
            raise NotImplementedError(self.aggr)
        return out

    def message(self, x_j):
        return x_j


class RGCNConv(MessagePassing):
    def __init__(self, in_channels, out_channels, num_relations, num_bases=None,
                 num_blocks=None, aggr="mean", root_weight=True, bias=True,
                 **kwargs):
        super().__init__(aggr=aggr, node_dim=0, **kwargs)
        if num_bases is not None or num_blocks is not None:
            raise NotImplementedError("shim covers the un-decomposed RGCNConv only")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_relations = num_relations
        self.num_bases = num_bases
        self.num_blocks = num_blocks
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.in_channels_l = in_channels[0]
        self.weight = Parameter(Tensor(num_relations, in_channels[0], out_channels))
        self.register_parameter("comp", None)
        if root_weight:
            self.root = Parameter(Tensor(in_channels[1], out_channels))
        else:
            self.register_parameter("root", None)
        if bias:
            self.bias = Parameter(Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.weight)
        glorot(self.comp)
        glorot(self.root)
        zeros(self.bias)
