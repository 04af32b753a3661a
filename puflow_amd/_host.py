"""Host-side hygiene shared by the entry points (no GPU work)."""
import os


def cpu_share() -> int:
    """CPUs this process may really use: the scheduler affinity capped by the cgroup's CFS quota (`cpu.max` = quota period).
    torch sizes its OpenMP pool by the VISIBLE CPUs (128 on the MI355X boxes) - in a container that owns 16 CPUs' worth of quota a
    pool of spinning threads exhausts the quota within a few ms and the whole process is throttled until the next 100 ms period,
    the thread that waits for the GPU included."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                tok = f.read().split()
            if path.endswith("cpu.max"):
                if tok[0] != "max":
                    n = min(n, max(1, int(tok[0]) // int(tok[1])))
            else:
                q = int(tok[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def limit_host_threads(cap: int = None) -> int:
    """torch intra-op threads := min(current, cpu_share() [, cap]); returns the previous setting."""
    import torch
    old = torch.get_num_threads()
    n = min(old, cpu_share())
    if cap is not None:
        n = min(n, cap)
    torch.set_num_threads(max(1, n))
    return old
