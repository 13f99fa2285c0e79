import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Worker threads of torch / numpy (set before they are imported): the GPU boxes show 256 hardware threads under a cgroup quota of
# 16 CPUs — 128 worker threads only get the whole process throttled (bench.py: usable_cpus).
if (os.cpu_count() or 1) > 16:
    for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(_k, "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/torchmd")
    skip_ref = pytest.mark.skip(reason="/root/reference not present (GPU box)")
    have_gpu = None
    for item in items:
        if "needs_reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
        if "gpu" in item.keywords:
            if have_gpu is None:
                import torch

                have_gpu = torch.cuda.is_available()
            if not have_gpu:
                item.add_marker(pytest.mark.skip(reason="needs a ROCm device (run through gpurun)"))


def pytest_report_header(config):
    """Which box this is: results that differ between boxes of a pool (clocks, partition mode, host) can then be told
    from results that differ between runs."""
    try:
        import torch

        if not torch.cuda.is_available():
            return None
        p = torch.cuda.get_device_properties(0)
        return (f"device: {p.name}, {p.multi_processor_count} CUs, {p.total_memory / 2**30:.0f} GiB, "
                f"arch {getattr(p, 'gcnArchName', '?')}, host cores {os.cpu_count()}")
    except Exception as exc:  # the header is information only
        return f"device: unknown ({exc})"
