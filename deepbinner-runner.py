#!/usr/bin/env python3
"""Run Deepbinner (MI355X build) from a source checkout: ``./deepbinner-runner.py classify ...``
— same role as the reference's ``deepbinner-runner.py:15-19``."""

from deepbinner_amd.deepbinner import main

if __name__ == '__main__':
    main()
