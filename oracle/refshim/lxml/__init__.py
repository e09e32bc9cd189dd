"""Stand-in: lxml.etree -> xml.etree.ElementTree (plain .xml only)."""
