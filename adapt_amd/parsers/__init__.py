from .xml_parser import scene_parsing  # noqa: F401
