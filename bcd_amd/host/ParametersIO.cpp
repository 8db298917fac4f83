// ParametersIO.cpp -- .bcd.json presets on a minimal flat-object JSON reader/writer (see ParametersIO.h).
#include "ParametersIO.h"
#include "Utils.h"

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

using namespace std;

namespace bcd
{

	namespace
	{
		struct JsonValue
		{
			enum Kind { e_string, e_number, e_bool, e_other } m_kind;
			string m_text; // string contents, or the literal for numbers / booleans
			double number() const { return atof(m_text.c_str()); }
			bool truth() const { return m_kind == e_bool ? m_text == "true" : number() != 0.; }
		};

		class FlatJsonParser
		{
		public:
			explicit FlatJsonParser(const string& text) : m_text(text), m_pos(0) {}

			bool parse(map<string, JsonValue>& o_rObject)
			{
				skipSpaces();
				if(!consume('{')) return false;
				skipSpaces();
				if(consume('}')) return true;
				while(true)
				{
					string key;
					skipSpaces();
					if(!parseString(key)) return false;
					skipSpaces();
					if(!consume(':')) return false;
					skipSpaces();
					JsonValue value;
					if(!parseValue(value)) return false;
					o_rObject[key] = value;
					skipSpaces();
					if(consume(',')) continue;
					return consume('}');
				}
			}

		private:
			void skipSpaces() { while(m_pos < m_text.size() && isspace(static_cast<unsigned char>(m_text[m_pos]))) ++m_pos; }
			bool consume(char c) { if(m_pos < m_text.size() && m_text[m_pos] == c) { ++m_pos; return true; } return false; }

			bool parseString(string& o_rString)
			{
				if(!consume('"')) return false;
				o_rString.clear();
				while(m_pos < m_text.size() && m_text[m_pos] != '"')
				{
					char c = m_text[m_pos++];
					if(c == '\\' && m_pos < m_text.size())
					{
						const char e = m_text[m_pos++];
						switch(e)
						{
							case 'n': c = '\n'; break;
							case 't': c = '\t'; break;
							case 'r': c = '\r'; break;
							case 'b': c = '\b'; break;
							case 'f': c = '\f'; break;
							case 'u': m_pos = min(m_pos + 4, m_text.size()); c = '?'; break; // non-ASCII escapes are not needed for these files
							default: c = e; // \" \\ \/
						}
					}
					o_rString.push_back(c);
				}
				return consume('"');
			}

			bool parseValue(JsonValue& o_rValue)
			{
				if(m_pos >= m_text.size()) return false;
				const char c = m_text[m_pos];
				if(c == '"') { o_rValue.m_kind = JsonValue::e_string; return parseString(o_rValue.m_text); }
				if(c == '{' || c == '[')
				{	// nested containers are skipped (none of the preset keys uses them)
					int depth = 0;
					bool inString = false;
					for(; m_pos < m_text.size(); ++m_pos)
					{
						const char d = m_text[m_pos];
						if(inString) { if(d == '\\') ++m_pos; else if(d == '"') inString = false; continue; }
						if(d == '"') inString = true;
						else if(d == '{' || d == '[') ++depth;
						else if(d == '}' || d == ']') { if(--depth == 0) { ++m_pos; break; } }
					}
					o_rValue.m_kind = JsonValue::e_other;
					return depth == 0;
				}
				const size_t start = m_pos;
				while(m_pos < m_text.size() && m_text[m_pos] != ',' && m_text[m_pos] != '}' && !isspace(static_cast<unsigned char>(m_text[m_pos]))) ++m_pos;
				o_rValue.m_text = m_text.substr(start, m_pos - start);
				if(o_rValue.m_text == "true" || o_rValue.m_text == "false") o_rValue.m_kind = JsonValue::e_bool;
				else if(o_rValue.m_text == "null") o_rValue.m_kind = JsonValue::e_other;
				else
				{
					char* end = nullptr;
					strtod(o_rValue.m_text.c_str(), &end);
					if(o_rValue.m_text.empty() || *end != '\0') return false;
					o_rValue.m_kind = JsonValue::e_number;
				}
				return true;
			}

			const string& m_text;
			size_t m_pos;
		};

		string quoted(const string& s)
		{
			string out = "\"";
			for(char c : s)
			{
				if(c == '"' || c == '\\') out.push_back('\\');
				out.push_back(c);
			}
			return out + "\"";
		}
	}

	bool ParametersIO::load(PipelineParameters& o_rParams, const string& i_rFilePath, PipelineParametersSelector i_selector)
	{
		if(i_rFilePath == "")
		{
			cerr << "Couldn't load parameters: empty file name" << endl;
			return false;
		}
		ifstream file(i_rFilePath);
		if(!file)
		{
			cerr << "Error: couldn't open file '" << i_rFilePath << "'" << endl;
			return false;
		}
		stringstream buffer;
		buffer << file.rdbuf();
		const string text = buffer.str();
		map<string, JsonValue> object;
		if(!FlatJsonParser(text).parse(object))
		{
			cerr << "Error: '" << i_rFilePath << "' is not a valid .bcd.json file" << endl;
			return false;
		}
		const string folderPath = Utils::extractFolderPath(i_rFilePath);
		auto find = [&](const char* key) -> const JsonValue* { auto it = object.find(key); return it == object.end() ? nullptr : &it->second; };
		const JsonValue* v;
		if(i_selector.m_inputFileNames)
		{
			if((v = find("inputColorFile"))) o_rParams.m_inputFileNames.m_colors = folderPath + v->m_text;
			if((v = find("inputHistoFile"))) o_rParams.m_inputFileNames.m_histograms = folderPath + v->m_text;
			if((v = find("inputCovarFile"))) o_rParams.m_inputFileNames.m_covariances = folderPath + v->m_text;
		}
		if(i_selector.m_prefilteringParameters)
		{
			if((v = find("performSpikeRemovalPrefiltering"))) o_rParams.m_prefilteringParameters.m_performSpikeRemoval = v->truth();
			if((v = find("spikeRemovalThresholdStDevFactor"))) o_rParams.m_prefilteringParameters.m_spikeRemovalThresholdStDevFactor = float(v->number());
		}
		if(i_selector.m_denoiserParameters)
		{
			DenoiserParameters& rParams = o_rParams.m_denoiserParameters.m_monoscaleParameters;
			if((v = find("nbOfScales"))) o_rParams.m_denoiserParameters.m_nbOfScales = int(v->number());
			if((v = find("histoDistanceThreshold"))) rParams.m_histogramDistanceThreshold = float(v->number());
			if((v = find("useCuda"))) rParams.m_useCuda = v->truth();
			if((v = find("nbOfCores"))) rParams.m_nbOfCores = int(v->number());
			if((v = find("patchRadius"))) rParams.m_patchRadius = int(v->number());
			if((v = find("searchWindowRadius"))) rParams.m_searchWindowRadius = int(v->number());
			if((v = find("randomPixelOrder"))) rParams.m_useRandomPixelOrder = v->truth();
			if((v = find("markedPixelsSkippingProbability"))) rParams.m_markedPixelsSkippingProbability = float(v->number());
			if((v = find("minEigenValue"))) rParams.m_minEigenValue = float(v->number());
		}
		return true;
	}

	bool ParametersIO::write(const PipelineParameters& i_rParams, const string& i_rFilePath, PipelineParametersSelector i_selector)
	{
		if(i_rFilePath == "")
		{
			cerr << "Couldn't save parameters: empty file name" << endl;
			return false;
		}
		ofstream file(i_rFilePath);
		if(!file)
		{
			cerr << "Error: couldn't write file '" << i_rFilePath << "'" << endl;
			return false;
		}
		const string folderPath = Utils::extractFolderPath(i_rFilePath);
		vector< pair<string, string> > entries;
		auto number = [](double d) { ostringstream oss; oss.precision(9); oss << d; return oss.str(); };
		auto boolean = [](bool b) { return string(b ? "true" : "false"); };
		if(i_selector.m_inputFileNames)
		{
			const InputFileNames& rNames = i_rParams.m_inputFileNames;
			entries.emplace_back("inputColorFile", quoted(Utils::getRelativePathFromFolder(rNames.m_colors, folderPath)));
			entries.emplace_back("inputHistoFile", quoted(Utils::getRelativePathFromFolder(rNames.m_histograms, folderPath)));
			entries.emplace_back("inputCovarFile", quoted(Utils::getRelativePathFromFolder(rNames.m_covariances, folderPath)));
		}
		if(i_selector.m_prefilteringParameters)
		{
			entries.emplace_back("performSpikeRemovalPrefiltering", boolean(i_rParams.m_prefilteringParameters.m_performSpikeRemoval));
			entries.emplace_back("spikeRemovalThresholdStDevFactor", number(i_rParams.m_prefilteringParameters.m_spikeRemovalThresholdStDevFactor));
		}
		if(i_selector.m_denoiserParameters)
		{
			const DenoiserParameters& rParams = i_rParams.m_denoiserParameters.m_monoscaleParameters;
			entries.emplace_back("nbOfScales", number(i_rParams.m_denoiserParameters.m_nbOfScales));
			entries.emplace_back("histoDistanceThreshold", number(rParams.m_histogramDistanceThreshold));
			entries.emplace_back("useCuda", boolean(rParams.m_useCuda));
			entries.emplace_back("nbOfCores", number(rParams.m_nbOfCores));
			entries.emplace_back("patchRadius", number(rParams.m_patchRadius));
			entries.emplace_back("searchWindowRadius", number(rParams.m_searchWindowRadius));
			entries.emplace_back("randomPixelOrder", boolean(rParams.m_useRandomPixelOrder));
			entries.emplace_back("markedPixelsSkippingProbability", number(rParams.m_markedPixelsSkippingProbability));
			entries.emplace_back("minEigenValue", number(rParams.m_minEigenValue));
		}
		file << "{" << endl;
		for(size_t i = 0; i < entries.size(); ++i)
			file << "    " << quoted(entries[i].first) << ": " << entries[i].second << (i + 1 < entries.size() ? "," : "") << endl;
		file << "}" << endl;
		return bool(file);
	}

} // namespace bcd
