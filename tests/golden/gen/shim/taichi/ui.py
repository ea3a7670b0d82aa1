"""import-only placeholder"""
