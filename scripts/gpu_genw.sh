cd $GRAFT_REPO_ROOT
for i in 1 2; do MULTI=0 timeout 200 python scripts/quick_all.py weather_sept_85 census1881 census-income 2>/dev/null; done
