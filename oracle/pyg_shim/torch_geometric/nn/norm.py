"""`torch_geometric.nn.norm.BatchNorm` = wrapper around BatchNorm1d held in
`.module`, hence the `.module.` state_dict keys."""
import torch


class BatchNorm(torch.nn.Module):
    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True,
                 track_running_stats=True):
        super().__init__()
        self.module = torch.nn.BatchNorm1d(in_channels, eps, momentum, affine,
                                           track_running_stats)
        self.reset_parameters()

    def reset_parameters(self):
        self.module.reset_parameters()

    def forward(self, x):
        return self.module(x)
