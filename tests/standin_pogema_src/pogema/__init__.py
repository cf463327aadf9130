"""STAND-IN source tree for tools/gen_golden_grid.py's rehearsal -- test infrastructure, NOT the reference.

Like the real package's `__init__`, this one cannot be imported without gymnasium: the grid-layer generator must get at
`grid_config.py` / `generator.py` / `grid.py` without executing it."""
raise ImportError("No module named 'gymnasium' (stand-in: the package __init__ must not be executed)")
