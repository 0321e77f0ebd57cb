from .main import ts2d_entry_point

ts2d_entry_point()
