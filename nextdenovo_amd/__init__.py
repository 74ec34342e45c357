
import os as _os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); the libraries' contexts keep 16+ streams
# busy, and streams that share a queue serialize (profiles/r03_hw_queues_ab.json: 12.1 s -> 5.7 s).  The libraries set the same
# default when they are loaded (csrc/capi.cpp); doing it here as well covers a process whose other packages (torch) touch HIP
# before the libraries are loaded.  A value the user has exported is kept.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
