"""``python -m gt4py_amd.distributed``: the self-check of the multi-GPU path on the devices of this job (selfcheck.run_selfcheck)."""

from .selfcheck import main

raise SystemExit(main())
