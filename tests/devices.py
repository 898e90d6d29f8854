"""`@pytest.mark.parametrize('device', DEVICES)`: the reference-pinned semantic fixtures run on the CPU (eager torch, the CPU suite) AND on
the MI355X (`cuda` leg, gpu mark: device tensors, the HIP kernels where the path has them) — same fixture, same tolerances."""
import pytest

DEVICES = [pytest.param('cpu', id='cpu'), pytest.param('cuda', id='cuda', marks=pytest.mark.gpu)]
