"""XML documents held by the reference's own tests (data fixtures), shared by the golden generator and
tests/test_builders.py. Test infrastructure only."""
# XML fixtures of the reference's own tests (data): tests/conftest.py:94-106 and tests/config_agents_from_xml_test.py:38-95
SIMPLE_NETWORK_XML = ('<network>  <links effectivecellsize="7.5">'
                      '    <link id="0" from="A" to="B" length="100" capacity="10" freespeed="10" permlanes="1"/>'
                      '    <link id="1" from="B" to="A" length="100" capacity="10" freespeed="10" permlanes="1"/>'
                      '  </links></network>')
EQUIL_NETWORK_XML = """<?xml version="1.0" encoding="utf-8"?>
<network name="equil test network">
   <nodes>
      <node id="1" x="-20000" y="0"/>
      <node id="2" x="-15000" y="0"/>
      <node id="3" x="-10000" y="0"/>
   </nodes>
   <links capperiod="01:00:00">
      <link id="1" from="1" to="2" length="25" capacity="1" freespeed="8.33" permlanes="1" />
      <link id="2" from="2" to="3" length="25" capacity="1" freespeed="8.33" permlanes="1" />
      <link id="3" from="3" to="1" length="25" capacity="1" freespeed="8.33" permlanes="1" />
   </links>
</network>
"""
EQUIL_POPULATION_XML = """<?xml version='1.0' encoding='utf-8'?>
<population>
  <person id="1">
    <plan>
      <act type="h" x="-20000" y="0" link="1" end_time="06:00" />
      <act type="w" x="-10000" y="0" link="3" end_time="07:00"/>
      <act type="h" x="-20000" y="0" link="1" end_time="08:00"/>
      <act type="w" x="-20000" y="0" link="3" end_time="09:00"/>
    </plan>
  </person>
  <person id="2">
    <plan>
      <act type="h" x="-20000" y="0" link="1" end_time="06:00" />
      <act type="w" x="-10000" y="0" link="3" end_time="07:00"/>
    </plan>
  </person>
  <person id="3">
    <plan>
    </plan>
  </person>
  <person id="4">
    <plan>
      <act type="h" x="-20000" y="0" link="3" end_time="06:30"/>
      <act type="w" x="-10000" y="0" link="1" end_time="07:00"/>
    </plan>
  </person>
</population>
"""
