"""No kernel touches a byte outside its operands: every GEMM kernel that accepts a descriptor (random draws, the skinny / decode
shapes in both rhs layouts, the benchmark's shapes), the reductions and copy_into run with each operand mapped on its own and
flush against unmapped address space -- at its end, then at its start (tools/guard_check.py; HIP virtual-memory API).  An access
one element out of bounds is a GPU memory fault, which kills the process: each pass therefore runs in a child, and the last line
the child printed names the launch at fault.  The detector itself is proven first (an output 64 bytes short MUST fault)."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _child(*args):
    return subprocess.run([sys.executable, str(ROOT / "tools" / "guard_check.py"), *args], capture_output=True, text=True, timeout=900)


def test_the_detector_faults_on_an_output_that_is_64_bytes_short():
    r = _child("--selftest")
    assert r.returncode != 0 and "THE DETECTOR DOES NOT WORK" not in r.stdout, r.stdout[-500:]
    assert "Memory access fault" in r.stderr, r.stderr[-500:]


@pytest.mark.parametrize("side", ["end", "front"])
@pytest.mark.parametrize("op", ["gemm", "reduce", "copy"])
def test_no_launch_touches_memory_outside_its_operands(op, side):
    r = _child("--ops", op, *(["--front"] if side == "front" else []))
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0 and lines and lines[-1] == "guard check complete", (lines[-2:], r.stderr[-600:])
