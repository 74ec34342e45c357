"""A CPU backend for nextdenovo_amd.stage.Shard built from oracle/ (test infrastructure): the overlap oracle
(oracle/mm_oracle.c) maps, the sort oracle (oracle/ovlsort_oracle.c) sorts.  Lets the sharding logic of the stage -- which
jobs a seed file needs, in which order the sort reads them, what a rank ends up correcting -- run on a box without a GPU."""
from __future__ import annotations

import numpy as np

import mm_util as M
import os_util as O
from nextdenovo_amd import overlap, ovl


class OracleBackend:
    def __init__(self, olib, preset_name):
        self.lib = M.bind(olib)
        O.bind(olib)
        self.preset_name = preset_name
        self.mid_occ = {}
        self.last_stats = None

    @staticmethod
    def _codes(rs):
        codes, off = ovl.unpack_codes(rs.words, rs.word_off, rs.lens)
        return (np.ascontiguousarray(rs.ids), np.ascontiguousarray(rs.lens), codes, off)

    def map(self, key, target, query, batch_size, dual):
        blob, mo = M.step1(self.lib, M.preset(self.preset_name, dual), self._codes(target), self._codes(query),
                           mid_occ=self.mid_occ.get(key, 0), batch_size=batch_size)
        self.mid_occ.setdefault(key, mo)   # one threshold per target file, as the device's index cache keeps it
        return overlap.from_decoded(ovl.decode_bytes(blob))

    def release(self, key, keep_stats=False):
        self.mid_occ.pop(key, None)

    def sort(self, files, seed_len, min_seed_len, k, flank):
        raw = [np.stack([f[n] for n in ("qname", "rev", "qs", "qe", "tname", "ts", "te", "match")], axis=1).astype(np.uint32)
               if f.size else np.zeros((0, 8), dtype=np.uint32) for f in files]
        _blob, bl, out = O.oracle_sort(self.lib, raw, np.ascontiguousarray(seed_len, dtype=np.uint32), min_seed_len, max_bin_cov=k, flank=flank)
        bl_list = [(int(x.split()[0]), x.split()[1]) for x in bl.splitlines()]
        return out.astype(overlap.REC), bl_list, {}
