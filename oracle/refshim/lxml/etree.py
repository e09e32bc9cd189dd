from xml.etree.ElementTree import *  # noqa: F401,F403
from xml.etree.ElementTree import parse, Comment  # noqa: F401


class _Comment:  # the reference only uses it in isinstance() checks
    pass
