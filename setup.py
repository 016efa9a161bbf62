#!/usr/bin/env python
"""pip install .  -- builds libmcevidence_hip.so with hipcc (gfx950) and installs the
mcevidence_amd package plus the drop-in MCEvidence module."""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class BuildWithHip(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "mcevidence_amd", "csrc"), "-j8"])
        super().run()


setup(
    name="mcevidence-amd",
    version="0.1.0",
    description="MI355X-native kNN Bayesian-evidence estimator (drop-in for MCEvidence)",
    packages=["mcevidence_amd"],
    py_modules=["MCEvidence"],
    # both libraries the Makefile builds: the HIP hot path and the host chain reader (chains.read_chain_file)
    package_data={"mcevidence_amd": ["libmcevidence_hip.so", "libmcechains.so"]},
    install_requires=["numpy"],
    cmdclass={"build_py": BuildWithHip},
    license="MIT",
)
