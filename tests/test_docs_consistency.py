"""DESIGN.md stays the CURRENT design: short, and its switch table knows every environment switch the package, bench.py and the C side read."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _switches():
    found = set()
    files = [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')]
    files += glob.glob(os.path.join(ROOT, 'offsetguided_amd', '**', '*.py'), recursive=True)
    files += glob.glob(os.path.join(ROOT, 'offsetguided_amd', 'csrc', '*'))
    for f in files:
        if os.path.isfile(f):
            found |= set(re.findall(r"""(?:environ\.get\(|environ\[|getenv\(|env_int\()\s*['"](OG_[A-Z0-9_]+)""", open(f, errors='replace').read()))
    return found


def test_design_is_short_and_lists_every_switch():
    design = open(os.path.join(ROOT, 'DESIGN.md')).read()
    assert design.count('\n') <= 300, 'DESIGN.md is the current design (<= 300 lines); narrative goes to EXPERIMENTS.md'
    switches = _switches()
    assert len(switches) >= 15, switches
    missing = sorted(s for s in switches if s not in design)
    assert not missing, f'environment switches read by the code but absent from DESIGN.md: {missing}'
