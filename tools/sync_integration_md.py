#!/usr/bin/env python
"""Re-quotes dpvo_amd/integration_stubs.py in INTEGRATION.md sections 1-3 (the four ```python blocks: common part, cuda_corr, cuda_ba,
lietorch_backends).  tests/test_capi.py::test_integration_md_quotes_the_stub_file fails when the two drift apart; run this after editing the stubs."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "dpvo_amd", "integration_stubs.py")).read()
ic, ib, il = src.index("class cuda_corr:"), src.index("class cuda_ba:"), src.index("class lietorch_backends:")
parts = [src[src.index("import ctypes"):ic].rstrip(), src[ic:ib].rstrip(), src[ib:il].rstrip(), src[il:].rstrip()]
path = os.path.join(ROOT, "INTEGRATION.md")
md = open(path).read()
a, b = md.index("## 1-3. The three import sites"), md.index("## 4. ")
sec = md[a:b]
it = iter(parts)
sec, n = re.subn(r"```python\n.*?```", lambda m: "```python\n" + next(it) + "\n```", sec, flags=re.S)
assert n == 4, n
open(path, "w").write(md[:a] + sec + md[b:])
print("INTEGRATION.md sections 1-3 re-quoted from dpvo_amd/integration_stubs.py")
