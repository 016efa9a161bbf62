#!/usr/bin/env python
"""Drop-in module name: ``from MCEvidence import MCEvidence`` and
``python MCEvidence.py <root> [flags]`` work as with the reference file of the same
name; everything lives in the ``mcevidence_amd`` package."""
from mcevidence_amd import (MCEvidence, MCSamples, cosmo_params_list, get_prior_volume,  # noqa: F401
                            iscosmo_param, params_info)

if __name__ == "__main__":
    import sys
    from mcevidence_amd.cli import main
    main(sys.argv[1:])
