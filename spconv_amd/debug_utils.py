"""``spconv.debug_utils`` (reference ``spconv/debug_utils.py:21-36``): the pickle dump of a failing
layer's inputs into ``SPCONV_DEBUG_SAVE_PATH``.  The layer modules call ``spconv_amd.tools.save_debug_data``;
this is the name user code knows."""
from spconv_amd.tools import save_debug_data as spconv_save_debug_data  # noqa: F401
