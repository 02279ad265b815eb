for ov in 4 0; do export PTMI355_OVERLAP=$ov; echo "lanes $ov"; bash profiles/tools/ab.sh "--config c4 --flags compact,bvh --steps 10 --warmup 2" mbA mbD mbC mbB; done
