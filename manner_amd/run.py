"""``python -m manner_amd.run <script.py> [args…]`` — run one of the reference's entry points (manner/train.py, manner/eval.py)
with the HIP operator classes installed, without editing a line of the reference:

    cd <checkout of andreeaiana/manner>
    PYTHONPATH=<this repository> python -m manner_amd.run manner/train.py experiment=cr_module_mind_title_scl_lf

``install()`` runs first (manner_amd/binding.py), then the script runs as ``__main__`` with ``sys.argv`` shifted, exactly as
``python <script.py> [args…]`` would (the script's directory is put first on ``sys.path`` as the interpreter does; the current
directory is added so that ``import manner`` finds the checkout — the reference's own scripts rely on ``pyrootutils`` for that,
reference manner/train.py:11)."""
from __future__ import annotations

import os
import runpy
import sys


def main() -> None:
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m manner_amd.run <script.py> [args...]")
    script = os.path.abspath(sys.argv[1])
    cwd = os.getcwd()
    if cwd not in sys.path:
        sys.path.insert(0, cwd)
    from manner_amd.binding import install
    rebound = install()
    print("[manner_amd] installed: " + "; ".join(f"{m.rsplit('.', 1)[1]}: {', '.join(v)}" for m, v in rebound.items()), file=sys.stderr)
    sys.argv = [script] + sys.argv[2:]
    sys.path.insert(0, os.path.dirname(script))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
