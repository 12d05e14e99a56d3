"""TEST INFRASTRUCTURE (build container only): fixtures that describe how the reference's training script
uses the mirrored API, for the machines where /root/reference does not exist (the GPU box).

    python oracle/gen_dropin_fixtures.py

    tests/golden/run_nerf_calls.json  every call `train()` of NeRF/run_nerf.py makes into the mirrored modules:
                                      [callee, n positional args, keyword names, has **kwargs]
    tests/golden/run_nerf_args.json   the argparse namespace NeRF/config_argparse.py produces for the tiny run of
                                      tests/dropin_support.train_argv (defaults of all 77 options included)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import dropin_support as S  # noqa: E402


def main():
    assert S.reference_available(), "needs /root/reference"
    out = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(S.REF_ROOT, "NeRF", "run_nerf.py")) as f:
        calls = S.call_surface(f.read(), "train")
    with open(os.path.join(out, "run_nerf_calls.json"), "w") as f:
        json.dump({"source": "NeRF/run_nerf.py::train", "calls": calls}, f, indent=1)
    mod = S.import_reference_run_nerf()
    args = mod.config_parser().parse_args(S.train_argv("BASEDIR", 4)[1:])
    with open(os.path.join(out, "run_nerf_args.json"), "w") as f:
        json.dump({"argv": S.train_argv("BASEDIR", 4)[1:], "namespace": vars(args)}, f, indent=1, sort_keys=True)
    print("wrote %d call records, %d options" % (len(calls), len(vars(args))))


if __name__ == "__main__":
    main()
