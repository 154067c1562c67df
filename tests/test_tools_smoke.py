"""The kept measurement / fixture tools (tools/*.py, tools/experiments/*.py) still fit the library: every one imports, and every
`lib.og_*` entry point a tool calls is either part of the C ABI (include/og_decoder.h via _lib.SIGNATURES, ABI v3) or one of the
diagnostic-build symbols (-DOG_*_STAMPS libraries built by tools/build_variants.sh / tools/build_stamps_lib.sh)."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = sorted(glob.glob(os.path.join(ROOT, 'tools', '*.py')) + glob.glob(os.path.join(ROOT, 'tools', 'experiments', '*.py')))
TOOLS = [t for t in TOOLS if not t.endswith('__init__.py')]
# symbols that exist only in diagnostic builds of the library (never in the product .so)
DIAGNOSTIC = {'og_k1_band_stamps', 'og_k1_wave_stamps', 'og_k1_debug_stamps', 'og_k3_debug_stamps', 'og_k3_wall_stamps',
              'og_conv3x3_debug_stamps', 'og_conv_band_debug_stamps', 'og_conv1x1_debug_stamps'}


def test_every_tool_imports():
    mods = [os.path.relpath(t, ROOT)[:-3].replace(os.sep, '.') for t in TOOLS]
    code = ("import importlib, sys\nsys.argv = ['x']\n"
            "for m in %r:\n    importlib.import_module(m)\nprint('imported', len(%r))\n" % (mods, mods))
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and f'imported {len(mods)}' in r.stdout, r.stderr[-2000:]


def test_tools_call_only_entry_points_of_the_abi():
    from offsetguided_amd import _lib
    assert _lib.ABI_VERSION == 3
    unknown = {}
    for t in TOOLS:
        src = open(t).read()
        names = set(re.findall(r"\blib\.(og_\w+)", src)) | set(re.findall(r"_lib\.lp\(lib, '(og_\w+)'", src))
        for n in names:
            if n in _lib.SIGNATURES or n in DIAGNOSTIC or n + '_bf16' in _lib.SIGNATURES:
                continue
            unknown.setdefault(os.path.basename(t), []).append(n)
    assert not unknown, f'tools call entry points the library does not export: {unknown}'
