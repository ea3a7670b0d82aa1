"""import-time stub for the absent `configargparse` (only needed if get_options() is called)."""
import argparse
class ArgumentParser(argparse.ArgumentParser):
    def add_argument(self, *a, **k):
        k.pop("is_config_file", None)
        return super().add_argument(*a, **k)
