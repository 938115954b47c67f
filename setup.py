from setuptools import setup

setup()   # (everything is in setup.cfg)
