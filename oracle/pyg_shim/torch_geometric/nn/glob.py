"""`GlobalAttention`: segment softmax with PyG's +1e-16."""
import torch
from .inits import reset


def _segment_softmax(src, index, num_nodes):
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    seg_max = torch.full((num_nodes,) + tuple(src.shape[1:]), float("-inf"),
                         dtype=src.dtype, device=src.device)
    seg_max = seg_max.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    out = (src - seg_max.index_select(0, index)).exp()
    seg_sum = torch.zeros((num_nodes,) + tuple(src.shape[1:]), dtype=src.dtype,
                          device=src.device).index_add_(0, index, out)
    return out / (seg_sum.index_select(0, index) + 1e-16)


class GlobalAttention(torch.nn.Module):
    def __init__(self, gate_nn, nn=None):
        super().__init__()
        self.gate_nn = gate_nn
        self.nn = nn
        self.reset_parameters()

    def reset_parameters(self):
        reset(self.gate_nn)
        reset(self.nn)

    def forward(self, x, batch, size=None):
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        size = int(batch[-1].item()) + 1 if size is None else size
        gate = self.gate_nn(x).view(-1, 1)
        x = self.nn(x) if self.nn is not None else x
        assert gate.dim() == x.dim() and gate.size(0) == x.size(0)
        gate = _segment_softmax(gate, batch, size)
        out = torch.zeros((size, x.size(1)), dtype=x.dtype, device=x.device)
        out = out.index_add(0, batch, gate * x)
        return out
