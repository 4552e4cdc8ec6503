"""Oracle helper (TEST INFRASTRUCTURE -- see oracle/__init__.py): a host copy of SOME rows of a device-resident
table, addressable by global row like the full array.  The 103,227,021-row code table of the full-size
workloads (13.2 GB) is generated on the device; the oracle only ever evaluates ``codes[rows]``, so the
checker pulls exactly the rows a test / bench block touches."""
import numpy as np


class HostRows:
    def __init__(self, table_dev, rows):
        import torch
        self.rows = np.unique(np.asarray(rows, dtype=np.int64))
        idx = torch.from_numpy(self.rows).to(table_dev.device)
        self.data = table_dev.index_select(0, idx).cpu().numpy()
        self.shape = tuple(table_dev.shape)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, rows):
        rows = np.asarray(rows, dtype=np.int64)
        pos = np.searchsorted(self.rows, rows)
        ok = np.array_equal(self.rows[np.minimum(pos, len(self.rows) - 1)], rows)
        assert ok, "row was not fetched to the host"
        return self.data[pos]
