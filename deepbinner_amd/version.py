"""Version of this implementation; tracks the reference release it is a drop-in for
(reference ``deepbinner/version.py:17`` -> 0.2.0)."""

__version__ = '0.2.0'
__backend__ = 'hip-gfx950'
