bash tools/evidence_round.sh 2>&1 | tail -30
