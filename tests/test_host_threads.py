"""Host hygiene (no GPU): the CPU share the entry points size torch's thread pool by, and the CLI's one-thread scope."""
import torch

from puflow_amd._host import cpu_share, limit_host_threads


def test_cpu_share_is_within_the_visible_cpus():
    import os
    n = cpu_share()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_limit_host_threads_never_raises_the_count_and_can_be_undone():
    old = torch.get_num_threads()
    try:
        prev = limit_host_threads(cap=2)
        assert prev == old and torch.get_num_threads() <= min(old, 2)
    finally:
        torch.set_num_threads(old)
    assert torch.get_num_threads() == old
