"""inverted_index_2_amd — MI355X-native posting-list engine (segment merge, intersection /
union, tombstone filter) behind the API of lezhnev74/inverted_index_2.  The compute lives in
libii2_hip.so (HIP, gfx950); see include/ii2.h for the C ABI and DESIGN.md for the layout."""
from .engine import Alignment, Context, DeviceArray, II2Error, Segment, Tombstones, comm_unique_id  # noqa: F401
