"""CPU (hipcc cross-compiles gfx950 without a GPU): properties of the generated ISA that the streaming key-addressing
kernel (mvin_amd/csrc/mvin_keyaddr_stream.hip) depends on and that a compiler change could silently break.

* M0 (the LDS base of its LDS-DMA instructions) is written exactly once per kernel and used by nothing else: the
  kernel's vmcnt accounting and its 8 KB LDS window assume no other M0 write, and every M0 write stalls on the DMA
  pieces in flight (profiles/r2/*gather_ceiling_dma_probe.txt).
* Every `s_waitcnt vmcnt` in it is one the source asked for (compile-time counts 0 / pieces-per-stage / 3): a
  compiler-inserted vmcnt(0) in front of the LDS reads -- what the LDS-DMA builtin gets -- would drain the stage in flight.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mvin_amd", "csrc")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


@pytest.fixture(scope="module")
def stream_isa(tmp_path_factory):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "keyaddr_stream.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
                    "-S", "--cuda-device-only", os.path.join(CSRC, "mvin_keyaddr_stream.hip"), "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_ZN4mvin22key_addr_stream_kernel\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        kernels[m.group(1)] = m.group(2)
    assert len(kernels) >= 20, "expected one body per (D, table dtype, padded memory count) instance"
    return kernels


def test_m0_written_once_and_used_by_nothing_else(stream_isa):
    for name, body in stream_isa.items():
        refs = [ln.strip() for ln in body.splitlines() if re.search(r"\bm0\b", ln)]
        assert len(refs) == 1 and refs[0].startswith("s_mov_b32 m0,"), (name, refs)


def test_only_the_requested_vmcnt_waits(stream_isa):
    for name, body in stream_isa.items():
        n_dma = len(re.findall(r"global_load_lds_dword", body))
        assert n_dma >= 10, name
        waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)]
        assert waits, name
        assert set(waits) <= {0, 1, 2, 3}, (name, sorted(set(waits)))
        # vmcnt(0): the h-set weight load, the top of the task loop, the two tail paths, the exit -- not one per LDS read
        assert waits.count(0) <= 8, (name, waits.count(0))
        assert sum(1 for w in waits if w in (1, 2)) >= 1, name
