bash scripts/r02_ab.sh
bash scripts/r02_tstamp1.sh
