"""Configuration: INI file -> nested attribute namespace, with a push/pop
context.  Same surface as the reference (Henbun/_settings.py:26-63,126-144;
henbunrc:6-14): `settings.numerics.jitter_level`, `get_settings()`,
`temp_settings(cfg)`; `henbunrc` is looked up in the package directory, then
the home directory, then the working directory (later files override).

Difference from the reference: dtype is NOT frozen at import (reference
param.py:26-27); it is read when a Model is created, and a Model can also be
given `dtype=` explicitly.
"""
from __future__ import annotations

import configparser
import copy
import os
from contextlib import contextmanager

import numpy as np


class _Namespace:
    def __init__(self, d=None):
        for k, v in (d or {}).items():
            setattr(self, k, v)

    def __repr__(self):
        return "Namespace(%s)" % ", ".join("%s=%r" % kv for kv in sorted(self.__dict__.items()))


def _parse(value: str):
    low = value.strip().lower()
    if low in ("true", "yes", "on"):
        return True
    if low in ("false", "no", "off"):
        return False
    for conv in (int, float):
        try:
            return conv(value)
        except ValueError:
            pass
    return value.strip()


def _load():
    here = os.path.dirname(os.path.abspath(__file__))
    cp = configparser.ConfigParser()
    paths = [os.path.join(here, "henbunrc"), os.path.join(os.path.expanduser("~"), ".henbunrc"),
             os.path.join(os.path.expanduser("~"), "henbunrc"), os.path.join(os.getcwd(), "henbunrc")]
    cp.read([p for p in paths if os.path.isfile(p)])
    return _Namespace({sec: _Namespace({k: _parse(v) for k, v in cp.items(sec)}) for sec in cp.sections()})


class _SettingsManager:
    """Attribute access goes to the settings on top of the stack."""

    def __init__(self, base):
        object.__setattr__(self, "_stack", [base])

    def __getattr__(self, name):
        return getattr(self._stack[-1], name)

    def __setattr__(self, name, value):
        setattr(self._stack[-1], name, value)

    def get_settings(self):
        """A deep copy of the current settings, to be edited and pushed with temp_settings."""
        return copy.deepcopy(self._stack[-1])

    @contextmanager
    def temp_settings(self, cfg):
        self._stack.append(cfg)
        try:
            yield
        finally:
            self._stack.pop()


settings = _SettingsManager(_load())


def np_float_type(name=None):
    name = name or settings.dtypes.float_type
    name = getattr(name, "__name__", None) or str(name)
    if "64" in name:
        return np.float64
    if "32" in name:
        return np.float32
    raise NotImplementedError("float_type must be float32 or float64, got %r" % (name,))
