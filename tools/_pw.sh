cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/ab_env.sh "persist2:OG_PW_PERSIST=2" "persist0:OG_PW_PERSIST=0" "persist2:OG_PW_PERSIST=2" "persist0:OG_PW_PERSIST=0" "persist2:OG_PW_PERSIST=2" "persist0:OG_PW_PERSIST=0"
