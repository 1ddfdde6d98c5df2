set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && rocprofv3 --list-avail 2>/dev/null | grep "Counter_Name" | sed 's/.*:\s*//' | sort -u > "$OLDPWD/gpurun_out/r06_counter_names.txt"; cd "$OLDPWD"
wc -l gpurun_out/r06_counter_names.txt
grep -E "^SQ_(WAIT|INSTS_LDS|LDS|ACTIVE_INST|INST_CYCLES|INSTS_VMEM|INSTS_VALU|INSTS_MFMA|BUSY|WAVE_CYC|INSTS_SALU|INSTS_SMEM|VALU_MFMA)|^TA_|^TCP_(PENDING|TCC_READ|TA_TCP|READ_TAG|GATE)|^LDS|^SQC_" gpurun_out/r06_counter_names.txt | tr '\n' ' '
