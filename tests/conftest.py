import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """The allowances of the parity checker that this session used (parity_cases.AUDIT), per context and model: written
    where the sweeps keep their evidence (gpurun_out/, copied to profiles/ by hand) -- C8_AUDIT_OUT names another file."""
    try:
        import json
        from parity_cases import AUDIT
    except Exception:
        return
    if not AUDIT.cases:
        return
    out = os.environ.get("C8_AUDIT_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_audit.json")
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        t = AUDIT.table()
        t["exitstatus"] = int(exitstatus)
        t["summary"] = {"cases": sum(t["cases"].values()), "allowances_used": sum(t["allowances_used"].values())}
        json.dump(t, open(out, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
