"""SASData: the three vectors the hot path reads, under the attribute names the reference uses
(dataobj/sasdata.py:51-75, dataobj/datavector.py:46-116).  File parsing, unit conversion, masking
and log-rebinning stay with the reference front-end (out of scope); `fromCsv` only
covers the plain `q; I; sigma` text layout of testdata/quickstartdemo1.csv for demos and tests."""
from __future__ import annotations

import numpy as np


class DataVector(object):
    def __init__(self, name, data, dataU=None):
        self.name = name
        self.binnedData = np.asarray(data, dtype=float)
        self.binnedDataU = None if dataU is None else np.asarray(dataU, dtype=float)
        self.sanitized = self.binnedData
        self.limit = [float(self.binnedData.min()), float(self.binnedData.max())] if len(self.binnedData) else [0., 0.]


class SASData(object):
    def __init__(self, q, intensity, sigma, f_limit=None, title="data"):
        self.title = title
        self.x0 = DataVector("q", q)
        self.f = DataVector("I", intensity, sigma)
        if f_limit is not None:          # limits of the UN-binned intensities (datavector.py:52)
            self.f.limit = [float(f_limit[0]), float(f_limit[1])]

    @property
    def q(self):
        return self.x0.binnedData

    @property
    def count(self):
        return len(self.x0.binnedData)

    def sphericalSizeEst(self):          # dataobj/sasdata.py:105 (pi / q limits)
        return np.array([np.pi / self.x0.limit[1], np.pi / self.x0.limit[0]])

    @classmethod
    def fromCsv(cls, filename, q_unit=1e9):
        raw = open(filename, "rb").read().decode("utf-8", "replace").replace("\r", "\n")
        rows = []
        for line in raw.split("\n"):
            parts = [x for x in line.replace(";", " ").replace(",", " ").split() if x]
            try:
                rows.append([float(x) for x in parts[:3]])
            except ValueError:
                continue
        arr = np.array([r for r in rows if len(r) == 3])
        return cls(arr[:, 0] * q_unit, arr[:, 1], arr[:, 2], title=filename)
