"""analysis helpers mirroring the reference's jupyter_utils (Hits, Histogram)"""
from .hits import Hits
from .histogram import Histogram
