"""`datasets` package shim (row f2): `datasets.ray_utils` resolves to the device version below, everything else of the
reference's package (generic_dataset.py, geo_utils.py, image_utils.py, ...) and the package-level names its own
`__init__.py` defines (`dataset_dict`, train.py:9) still come from the reference checkout further down sys.path."""
import os
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)

from _objnerf_dropin import run_reference_init  # noqa: E402  (dropin/ is on sys.path: that is how this package was found)

run_reference_init(globals(), os.path.dirname(os.path.abspath(__file__)))
