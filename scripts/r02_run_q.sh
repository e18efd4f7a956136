bash scripts/r02_ab.sh
BALS=2 bash scripts/r02_tstamp2.sh
