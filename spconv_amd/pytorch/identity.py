"""``spconv.pytorch.identity.Identity``: pass-through layer that also answers the
``input_spatial_size`` query of the reference's shape-inference helpers
(reference ``spconv/pytorch/identity.py``)."""
import torch


class Identity(torch.nn.Identity):
    @staticmethod
    def input_spatial_size(out_size):
        """A pass-through layer needs an input as large as its output."""
        return out_size
