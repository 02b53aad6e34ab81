"""The ``_C`` replacement a maintainer of the reference would add -- loaded VERBATIM from INTEGRATION.md section 2.

INTEGRATION.md shows the ctypes stubs that stand in for the reference's pybind module (RAST/ext.cpp:15-20). So that the document
cannot rot, this module does not restate them: it extracts the python code blocks of section 2 from INTEGRATION.md and executes
them (only the library path is made absolute). tests/test_boundary_gpu.py calls the resulting functions with the argument tuples
of RAST/diff_gof_rasterization/__init__.py:61-84 and :269-293."""
import os
import re

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(_ROOT, "f3d-gaus_amd", "csrc", "libf3dg_hip.so")


def _section2_code():
    text = open(os.path.join(_ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2."):text.index("## 3.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) >= 2, "INTEGRATION.md section 2 lost its code blocks"
    return "\n".join(blocks).replace('C.CDLL("libf3dg_hip.so")', "C.CDLL(%r)" % _LIB)


SOURCE = _section2_code()
exec(compile(SOURCE, "INTEGRATION.md#2", "exec"), globals())
