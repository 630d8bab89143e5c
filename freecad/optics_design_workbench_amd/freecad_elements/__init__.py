"""host-side mirrors of the reference's document element proxies"""
from . import point_source
