"""`utils` package shim (row f2): `utils.bbox_utils` resolves to the device version below; the rest of the reference's
package (util.py, metrics.py, train_helper.py) and the names its own `__init__.py` defines (`get_optimizer`,
`get_scheduler`, `get_parameters`, ...: train.py:22) still come from the reference checkout further down sys.path."""
import os
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)

from _objnerf_dropin import run_reference_init  # noqa: E402

run_reference_init(globals(), os.path.dirname(os.path.abspath(__file__)))
