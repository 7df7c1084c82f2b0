"""Stand-in for the `future` py2/3 compatibility package (test infrastructure, see README.md)."""
