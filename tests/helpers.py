"""Small deterministic pieces shared by tools/make_golden.py (which runs the reference's code on them) and the tests
(which run this build's code on the same things)."""
import torch


def search_toy_network():
    """Stand-in network of the search loop: [B, 2, 2048] -> softmax over 2 classes (CPU / any device, fp32)."""
    g = torch.Generator().manual_seed(1234)
    w = torch.randn(2 * 2048, 2, generator=g) * 0.05

    class Net(torch.nn.Module):
        def forward(self, x):
            return torch.softmax(x.reshape(x.shape[0], -1) @ w.to(x.device), dim=1)
    return Net()
