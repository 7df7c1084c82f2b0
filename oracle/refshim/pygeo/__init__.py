"""Stand-in for `pygeo` (SEG-Y reader) -- only needs to import (test infrastructure)."""
