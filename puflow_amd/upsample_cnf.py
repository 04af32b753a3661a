"""Inference CLI of the continuous (CNF) model - the reference's `modules/continuous/upsample.py` (identical to the
discrete script except for the model class, `continuous/upsample.py:15,29`):

  python -m puflow_amd.upsample_cnf --source=in/ --target=out/ --checkpoint=puflow-x4-cnf-pu1k.pt --up_ratio=4
"""
from __future__ import annotations

from . import upsample
from .cnf import PointInterpFlow


def main(argv=None):
    upsample.main(argv, network_cls=PointInterpFlow)


if __name__ == "__main__":
    main()
