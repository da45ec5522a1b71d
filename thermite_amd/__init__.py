"""thermite_amd -- MI355X-native seed-and-extend hot path of the thermite RNA aligner.

The product is the shared library built from thermite_amd/csrc (C ABI: include/thermite.h and
include/thermite_io.h).  This package holds the Python host side used by the tests and the
benchmark: `capi` (ctypes binding), `refdata` (FASTA/GTF/FASTQ restatement in Python), `synth`
(synthetic workloads), `sharding` (read shards per rank) and `validate` (sequence-level checker).
"""
__version__ = "0.1"
