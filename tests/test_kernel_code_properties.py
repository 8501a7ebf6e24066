"""Properties of the gfx950 code hipcc emits for the shipped kernels (cross-compiled here, no GPU needed):

* no kernel asks for the AQL dispatch packet or the queue pointer.  The LM step kernel once did: a private array the
  compiler promoted to LDS was indexed by a flat thread id computed from the workgroup's shape, which it loaded from the
  dispatch packet -- a scalar load from the queue's ring buffer (host-side memory) on the one lane every LM iteration waits
  for, 1.2 us per iteration (profiles/r02_lm_step_stamps.txt);
* no kernel spills to scratch (private segment size 0): the evaluation kernels' register budgets are part of their design
  (DESIGN section 4), a spill would turn arithmetic into memory round trips silently."""
import os
import re
import subprocess

import pytest

from edge_alignment_amd import build

CSRC = os.path.dirname(build.LIB).replace("lib", "csrc")


@pytest.mark.parametrize("source,extra", build.SOURCES[:3])   # (ea_capi.hip holds no kernels)
def test_no_kernel_reads_the_dispatch_packet_or_spills(tmp_path, source, extra):
    src = os.path.join(os.path.dirname(os.path.dirname(build.LIB)), source)
    out = str(tmp_path / "k.s")
    cmd = [build._hipcc()] + [f for f in build.FLAGS if f != "-fPIC"] + list(extra) + ["-S", "--cuda-device-only", "-o", out, src]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S)
    assert len(kernels) >= 8, source
    for name, body in kernels:
        for key in ("amdhsa_user_sgpr_dispatch_ptr", "amdhsa_user_sgpr_queue_ptr"):
            m = re.search(key + r" (\d+)", body)
            assert m and m.group(1) == "0", (name, key)
        m = re.search(r"amdhsa_private_segment_fixed_size (\d+)", body)
        assert m and m.group(1) == "0", (name, "scratch bytes " + m.group(1))
