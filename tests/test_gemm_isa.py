"""Compiler regression guard for the LDS-DMA GEMM pipeline (csrc/kernels/gemm.hip, gemm_pipeline_dma4).

The pipeline's correctness AND speed rest on where the compiler puts waits: an LDS read of a DMA-staged buffer is ordered
only by the issuing waves' counted `s_waitcnt vmcnt(N)` followed by a barrier, and one compiler-inserted
`s_waitcnt vmcnt(0)` in front of the fragment reads collapses the prefetch depth to zero (measured: 10-15 % slower than
register staging; that is why the transposed reads are inline asm).  This test compiles the file for gfx950 (no GPU needed)
and checks, in every kernel that contains LDS-DMA loads, that inside the DMA region
  * every vmcnt wait is one of the pipeline's own (none inserted between a DMA and the LDS reads of its phase),
  * a wait is followed by its barrier before any further LDS read or DMA,
  * the steady-state phase has the shape barrier, DMAs, reads, wait, barrier."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
OWN_WAITS = {"vmcnt(4)", "vmcnt(0)"}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_lds_dma_pipeline_waits_are_exactly_the_pipelines_own(tmp_path):
    out = tmp_path / "gemm.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-I",
                    os.path.join(ROOT, "csrc", "kernels"), os.path.join(ROOT, "csrc", "kernels", "gemm.hip"), "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    text = out.read_text()
    kernels = re.split(r"\n(?=_ZN3dtf[^\n]*:\s*;\s*@)", text)
    checked = 0
    for body in kernels:
        name = body.split(":", 1)[0]
        lines = [l.strip() for l in body.split("\n")]
        lines = lines[:next((i for i, l in enumerate(lines) if l.startswith("s_endpgm")), len(lines))]
        dma = [i for i, l in enumerate(lines) if l.startswith("buffer_load_dwordx4") and l.endswith("lds")]
        if not dma:
            continue
        checked += 1
        region = lines[dma[0]:dma[-1] + 40]     # the main loop and its last wait / barrier
        events = []
        for l in region:
            if l.startswith("s_barrier"):
                events.append("B")
            elif l.startswith("buffer_load_dwordx4") and l.endswith("lds"):
                events.append("D")
            elif l.startswith("ds_read"):
                events.append("R")
            elif l.startswith("s_waitcnt") and "vmcnt" in l:
                w = l[len("s_waitcnt"):].strip()
                assert w in OWN_WAITS, f"{name}: compiler-inserted wait '{l}' inside the DMA pipeline"
                events.append("W")
        s = "".join(events)
        assert "W" in s and "B" in s, (name, s)
        # after a wait, the next pipeline event is the barrier (no LDS read, no DMA in between)
        for m in re.finditer("W", s):
            nxt = s[m.end():m.end() + 1]
            assert nxt in ("B", "W", ""), f"{name}: '{nxt}' between the vmcnt wait and its barrier: ...{s[max(0, m.start() - 12):m.end() + 6]}"
        # every phase of the steady state both stages and reads: B D.. R.. W B
        assert re.search(r"BD+R+W+B", s), (name, s)
    assert checked >= 6, f"expected the NT (3 epilogues + fused head wgrad), TN (3) and NN DMA kernels, found {checked}"
