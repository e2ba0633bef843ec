"""Import alias for the package directory ``image-editing-framework_amd/``.

The directory name carries a hyphen (it mirrors the reference repo's name), which the
``import`` statement cannot spell.  ``import ief_amd`` loads that directory as a regular
package under the name ``ief_amd`` so that ``ief_amd.unet``, ``ief_amd.p2p.model.register``
... resolve to files inside it.  Nothing else lives here.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image-editing-framework_amd")


def _load():
    name = "ief_amd"
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod  # replaces this shim: later ``import ief_amd.x`` sees the package
    spec.loader.exec_module(mod)
    return mod


_load()
