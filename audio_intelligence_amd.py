"""Import alias: the package directory is `audio-intelligence_amd/` (a hyphen is not importable),
so `import audio_intelligence_amd` resolves here and adopts that directory as its package path."""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "audio-intelligence_amd")
__path__ = [_pkg_dir]
__file__ = _os.path.join(_pkg_dir, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
