"""mirge_amd: MI355X-native engine for miRge2.0's annotate-mode hot path
(the bowtie cascade + count tally).  See DESIGN.md."""
__version__ = "0.1.0"
